#!/bin/bash
# On the GPU box: a longer run of two generators over fresh seeds -> gpurun_out/fuzz3_long_<part>.txt     bash scratch/fuzz_round3_long.sh a|b
part=$1
mkdir -p gpurun_out
out=gpurun_out/fuzz3_long_$part.txt
: > $out
run() { echo "== scratch/$1" >> $out; timeout -k 10 $4 python scratch/$1 $2 $3 > gpurun_out/fuzz3_long_$1.log 2>&1; rc=$?; tail -1 gpurun_out/fuzz3_long_$1.log >> $out; echo "$1 rc=$rc"; [ $rc -eq 0 ] || exit 1; }
case $part in
  a) run fuzz_many.py 402000 407000 1100 ;;
  d) run fuzz_many.py 407000 437000 1150 ;;
  e) run fuzz_degenerate.py 95500 110500 1150 ;;
  b) run fuzz_degenerate.py 91500 95500 1100 ;;
  c) run fuzz_mixed.py 30500 36000 1150 ;;
  f) run fuzz_extremes.py 9200 10700 380 && run fuzz_camera.py 9200 10700 380 && run fuzz_meshes.py 9200 10700 380 ;;
  g) run fuzz_grid.py 6100 6900 500 && run fuzz_counts.py 8100 8500 300 && run fuzz_shards.py 9300 10800 380 ;;
  h) run fuzz_depth_map.py 9200 10700 380 && run fuzz_refractive_meshes.py 9200 10000 500 ;;
  i) run fuzz_odd_materials.py 1000 4000 560 && run fuzz_far_camera.py 1000 3000 560 ;;
  j) run fuzz_far_camera.py 3000 5000 1100 ;;
  k) run fuzz_mixed.py 36200 42000 1150 ;;
  l) run fuzz_mixed.py 42000 48000 1150 ;;
  m) run fuzz_mixed.py 48000 54000 1150 ;;
  n) run fuzz_odd_materials.py 4000 10000 1150 ;;
  o) run fuzz_far_camera.py 6000 10000 1150 ;;
esac
cat $out
