#!/bin/bash
# On the GPU box: a long run of every fuzz generator over fresh seeds, last line of each -> gpurun_out/fuzz_long.txt
out=gpurun_out/fuzz_long.txt
: > $out
run() { echo "== scratch/$1" >> $out; timeout -k 10 $4 python scratch/$1 $2 $3 > gpurun_out/fuzzlong_$1.log 2>&1; rc=$?; tail -1 gpurun_out/fuzzlong_$1.log >> $out; echo "$1 rc=$rc"; [ $rc -eq 0 ] || exit 1; }
run fuzz_many.py 300000 330000 900 && run fuzz_degenerate.py 70000 85000 900 && run fuzz_mixed.py 20000 24000 900 && run fuzz_counts.py 7000 7400 600 && run fuzz_grid.py 5000 5800 900 && run fuzz_shards.py 7000 8500 900 && run fuzz_meshes.py 7000 8500 600 && run fuzz_extremes.py 7000 8500 600 && run fuzz_camera.py 7000 8500 600 && run fuzz_depth_map.py 7000 8500 600 && run fuzz_refractive_meshes.py 7000 7800 600
