#!/bin/bash
# the whole GPU suite, then the default bench command
mkdir -p gpurun_out/suite
timeout -k 10 1000 python -u -m pytest tests -x -q --timeout 600 --timeout-method thread -m gpu > gpurun_out/suite/tests.log 2>&1
rc=$?
tail -4 gpurun_out/suite/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/suite/bench_default.json 2> gpurun_out/suite/bench_default.err
rc=$?
[ $rc -eq 0 ] || { tail -5 gpurun_out/suite/bench_default.err; exit $rc; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/suite/bench_default.json"))
print(round(d["value"]), "Mrays/s", round(d["ms_per_step"], 4), "ms/step", d["config"]["views_per_pass"], "views", {k: round(v, 3) for k, v in d["roofline"]["kernels_ms_per_frame"].items()},
      "seq", {k: round(v, 3) for k, v in d["roofline"]["one_frame_at_a_time"]["kernels_ms_per_frame"].items()}, "list bytes per view", d["config"]["list_bytes_per_view"], "roofline frac", d["roofline"].get("frac"), "cpu", d["cpu_baseline"]["value"])
PY
