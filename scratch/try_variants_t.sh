#!/bin/bash
# On the GPU box: per-kernel times of the library variants under scratch/variants/ WITHOUT the smoke test (for probes that give
# wrong pictures on purpose): one frame at a time and 16 views per pass
for v in scratch/variants/*.so; do
  cp $v rust_raytracer_amd/librt_hip.so
  echo "== $v"
  timeout -k 10 200 python scratch/kernel_times.py reflective_bunnies_1080p 2>/dev/null | sed -n 1p
  VIEWS=16 timeout -k 10 200 python scratch/kernel_times.py reflective_bunnies_1080p 2>/dev/null | sed -n 1p
done
