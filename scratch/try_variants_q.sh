#!/bin/bash
# On the GPU box: every library variant under scratch/variants/ in turn: smoke test against the oracle, then per-kernel times per
# frame one frame at a time (whole frame, 1/8 shard) and queued (8 and 16 views per pass).  bash scratch/try_variants_q.sh [workload]
W=${1:-reflective_bunnies_1080p}
for v in scratch/variants/*.so; do
  cp $v rust_raytracer_amd/librt_hip.so
  echo "== $v"
  timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids || exit 1
  timeout -k 10 200 python scratch/kernel_times.py $W 2>/dev/null || exit 1
  for n in ${VIEWS_LIST:-8 16}; do VIEWS=$n timeout -k 10 200 python scratch/kernel_times.py $W 2>/dev/null | sed -n 1p | sed "s/^/views $n: /"; done
done
