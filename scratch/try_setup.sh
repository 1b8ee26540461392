#!/bin/bash
mkdir -p gpurun_out/variants
cp rust_raytracer_amd/librt_hip.so /tmp/keep.so
for v in scratch/variants/*.so; do
  cp $v rust_raytracer_amd/librt_hip.so
  echo "== $v"
  timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids || exit 1
  VIEWS=16 timeout -k 10 200 python scratch/kernel_times.py reflective_bunnies_1080p 2>/dev/null | head -1 | sed "s/^/views 16: /"
  timeout -k 10 300 python scratch/kernel_times.py reflective_bunnies_1080p instance_grid_2p6m_4k 2>/dev/null | grep "1/1"
  VIEWS=8 timeout -k 10 300 python scratch/kernel_times.py instance_grid_2p6m_4k 2>/dev/null | head -1 | sed "s/^/views 8: /"
done
cp /tmp/keep.so rust_raytracer_amd/librt_hip.so
