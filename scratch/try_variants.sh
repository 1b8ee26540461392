#!/bin/bash
# On the GPU box: per-kernel times of the library variants under scratch/variants/ (each replaces the in-tree library in the box's copy)
set -e
for v in scratch/variants/*.so; do
  cp $v rust_raytracer_amd/librt_hip.so
  echo "== $v"
  timeout -k 10 300 python scratch/kernel_times.py ${WORKLOADS:-reflective_bunnies_1080p} 2>/dev/null
  VIEWS=8 timeout -k 10 120 python scratch/kernel_times.py reflective_bunnies_1080p 2>/dev/null
done
