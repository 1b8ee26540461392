#!/bin/bash
# On the GPU box: per-kernel times with 4 and 5 list blocks per compute unit
for b in 5 4; do
  echo "== RT_LIST_BLOCKS_PER_CU=$b"
  RT_LIST_BLOCKS_PER_CU=$b timeout -k 10 200 python scratch/kernel_times.py reflective_bunnies_1080p 2>/dev/null || exit 1
  RT_LIST_BLOCKS_PER_CU=$b VIEWS=8 timeout -k 10 200 python scratch/kernel_times.py reflective_bunnies_1080p 2>/dev/null || exit 1
done
