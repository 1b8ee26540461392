#!/bin/bash
# On the GPU box: every fuzz generator over fresh seeds, last line of each -> gpurun_out/fuzz_round.txt
out=gpurun_out/fuzz_round.txt
: > $out
run() { echo "== scratch/$1" >> $out; timeout -k 10 $4 python scratch/$1 $2 $3 > gpurun_out/fuzz_$1.log 2>&1; rc=$?; tail -1 gpurun_out/fuzz_$1.log >> $out; echo "$1 rc=$rc"; [ $rc -eq 0 ] || exit 1; }
run fuzz_many.py 200000 202000 500 && run fuzz_degenerate.py 60000 61500 500 && run fuzz_mixed.py 12000 12500 400 && run fuzz_counts.py 6000 6100 400 && run fuzz_grid.py 4000 4100 400 && run fuzz_shards.py 6000 6300 400 && run fuzz_meshes.py 6000 6200 300 && run fuzz_extremes.py 6000 6200 300 && run fuzz_camera.py 6000 6200 300 && run fuzz_depth_map.py 6000 6200 300
