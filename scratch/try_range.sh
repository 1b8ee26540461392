#!/bin/bash
# the 64-instance grid with the range tree over instances (default) and with the flat scan of culling records in LDS
for m in 16 64; do
  echo "== RT_RANGE_TREE_MIN=$m"
  RT_RANGE_TREE_MIN=$m timeout -k 10 300 python scratch/kernel_times.py instance_grid_2p6m_4k 2>/dev/null | grep "1/1"
  RT_RANGE_TREE_MIN=$m VIEWS=8 timeout -k 10 300 python scratch/kernel_times.py instance_grid_2p6m_4k 2>/dev/null | head -1 | sed "s/^/views 8: /"
  RT_RANGE_TREE_MIN=$m RT_FUSE_SETUP=0 VIEWS=8 timeout -k 10 300 python scratch/kernel_times.py instance_grid_2p6m_4k 2>/dev/null | head -1 | sed "s/^/views 8, set-up as a kernel: /"
done
