#!/bin/bash
# On the GPU box: per-kernel times at 8 views per pass of the library variants under scratch/variants/
set -e
for v in scratch/variants/*.so; do
  cp $v rust_raytracer_amd/librt_hip.so
  echo "== $v"
  VIEWS=8 timeout -k 10 120 python scratch/kernel_times.py ${WORKLOADS:-reflective_bunnies_1080p} 2>/dev/null | head -1
done
