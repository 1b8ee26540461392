#!/bin/bash
mkdir -p gpurun_out/variants
cp rust_raytracer_amd/librt_hip.so /tmp/keep.so
for cfg in "a_base 6" "b_shade7 7"; do
  set -- $cfg
  cp scratch/variants/$1.so rust_raytracer_amd/librt_hip.so
  echo "== $1, $2 list blocks per CU"
  RT_LIST_BLOCKS_PER_CU=$2 timeout -k 10 200 python scratch/kernel_times.py reflective_bunnies_1080p 2>/dev/null | head -1
  RT_LIST_BLOCKS_PER_CU=$2 VIEWS=16 timeout -k 10 200 python scratch/kernel_times.py reflective_bunnies_1080p 2>/dev/null | head -1 | sed "s/^/views 16: /"
done
cp /tmp/keep.so rust_raytracer_amd/librt_hip.so
