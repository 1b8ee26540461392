#!/bin/bash
mkdir -p gpurun_out/moot
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -m gpu -k "nobody_needs or api_errors or sized_by" > gpurun_out/moot/tests.log 2>&1
rc=$?
tail -5 gpurun_out/moot/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --workload instance_grid_2p6m_4k > gpurun_out/moot/bench_grid_2p6m.json 2> gpurun_out/moot/bench_grid_2p6m.err
rc=$?
[ $rc -eq 0 ] || { tail -5 gpurun_out/moot/bench_grid_2p6m.err; exit $rc; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/moot/bench_grid_2p6m.json"))
print(round(d["value"]), "Mrays/s", round(d["ms_per_step"], 4), "ms/step", {k: round(v, 3) for k, v in d["roofline"]["kernels_ms_per_frame"].items()},
      "seq", {k: round(v, 3) for k, v in d["roofline"]["one_frame_at_a_time"]["kernels_ms_per_frame"].items()}, d["config"].get("shadow_not_traced"), d["config"]["shadow"])
PY
