#!/bin/bash
# bench.py for every workload: bash scratch/bench_all.sh <tag>  (output: gpurun_out/<tag>/bench_<workload>.json)
TAG=$1
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
for w in reflective_bunnies_1080p reflective_bunnies_1080p_sphere_standin cornell_box_720p mirror_sphere_1080p_depth16 infinity_santa_1080p_depth16 instance_grid_2p6m_4k instance_grid_10m_4k; do
  timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err || echo "$w FAILED"
  echo "$w done"
done
