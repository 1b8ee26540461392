#!/bin/bash
# On the GPU box: a lone frame (and a lone 1/8 shard) with rt_tail from depth N on (0 = never)
for t in 0 3 2 4 1; do
  echo "== RT_TAIL_FIRST=$t"
  RT_TAIL_FIRST=$t timeout -k 10 200 python scratch/kernel_times.py ${WORKLOADS:-reflective_bunnies_1080p} 2>/dev/null || exit 1
done
