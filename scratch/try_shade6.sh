#!/bin/bash
# On the GPU box: rt_shade at five and at six waves per SIMD, with five and six list blocks per compute unit
mkdir -p gpurun_out/variants
cp rust_raytracer_amd/librt_hip.so /tmp/keep.so
for v in a_base b_shade6; do
  cp scratch/variants/$v.so rust_raytracer_amd/librt_hip.so
  for bpc in 5 6; do
    echo "== $v, $bpc list blocks per CU"
    RT_LIST_BLOCKS_PER_CU=$bpc timeout -k 10 200 python scratch/kernel_times.py reflective_bunnies_1080p 2>/dev/null | head -1
    RT_LIST_BLOCKS_PER_CU=$bpc VIEWS=16 timeout -k 10 200 python scratch/kernel_times.py reflective_bunnies_1080p 2>/dev/null | head -1 | sed "s/^/views 16: /"
  done
done
cp /tmp/keep.so rust_raytracer_amd/librt_hip.so
