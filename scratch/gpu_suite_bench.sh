#!/bin/bash
# the whole GPU suite, then the headline bench at 48 and 20 steps
mkdir -p gpurun_out/suite
timeout -k 10 1000 python -u -m pytest tests -x -q --timeout 600 --timeout-method thread -m gpu > gpurun_out/suite/tests.log 2>&1
rc=$?
tail -8 gpurun_out/suite/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 48 --warmup 16 > gpurun_out/suite/bench48.json 2> gpurun_out/suite/bench48.err
rc=$?
[ $rc -eq 0 ] || { tail -5 gpurun_out/suite/bench48.err; exit $rc; }
timeout -k 10 300 python bench.py > gpurun_out/suite/bench_default.json 2> gpurun_out/suite/bench_default.err
rc=$?
python - <<'PY'
import json
for n in ("bench48", "bench_default"):
    d = json.load(open("gpurun_out/suite/%s.json" % n))
    print(n, round(d["value"]), "Mrays/s", round(d["ms_per_step"], 4), "ms/step", {k: round(v, 3) for k, v in d["roofline"]["kernels_ms_per_frame"].items()},
          "seq", {k: round(v, 3) for k, v in d["roofline"]["one_frame_at_a_time"]["kernels_ms_per_frame"].items()})
PY
exit $rc
