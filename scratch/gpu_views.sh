#!/bin/bash
mkdir -p gpurun_out/views
for cfg in "16 48" "32 96" "24 72"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --views-per-pass $1 --steps $2 --warmup $1 > gpurun_out/views/bench_v$1.json 2> gpurun_out/views/bench_v$1.err || { tail -3 gpurun_out/views/bench_v$1.err; exit 1; }
  python - $1 <<'PY'
import json, sys
d = json.load(open("gpurun_out/views/bench_v%s.json" % sys.argv[1]))
print("views", sys.argv[1], round(d["value"]), "Mrays/s", round(d["ms_per_step"], 4), "ms/step", {k: round(v, 3) for k, v in d["roofline"]["kernels_ms_per_frame"].items()}, d["config"]["list_bytes_per_view"])
PY
done
