#!/bin/bash
# On the GPU box: a part of the fuzz generators over fresh seeds, last line of each -> gpurun_out/fuzz3_<part>.txt
#   bash scratch/fuzz_round3.sh a|b|c
part=$1
mkdir -p gpurun_out
out=gpurun_out/fuzz3_$part.txt
: > $out
run() { echo "== scratch/$1" >> $out; timeout -k 10 $4 python scratch/$1 $2 $3 > gpurun_out/fuzz3_$1.log 2>&1; rc=$?; tail -1 gpurun_out/fuzz3_$1.log >> $out; echo "$1 rc=$rc"; [ $rc -eq 0 ] || exit 1; }
case $part in
  a) run fuzz_many.py 400000 402000 520 && run fuzz_degenerate.py 90000 91500 520 ;;
  b) run fuzz_mixed.py 30000 30500 400 && run fuzz_counts.py 8000 8100 250 && run fuzz_grid.py 6000 6100 250 && run fuzz_shards.py 9000 9300 250 ;;
  c) run fuzz_meshes.py 9000 9200 220 && run fuzz_extremes.py 9000 9200 220 && run fuzz_camera.py 9000 9200 220 && run fuzz_depth_map.py 9000 9200 200 && run fuzz_refractive_meshes.py 9000 9200 220 ;;
esac
cat $out
