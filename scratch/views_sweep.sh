#!/bin/bash
# On the GPU box: the headline bench line for several (views per pass, passes in flight)
for cfg in "8 3" "16 2" "16 3" "32 1" "32 2" "4 6"; do
  set -- $cfg
  echo "== views-per-pass $1 frames-in-flight $2"
  timeout -k 10 200 python bench.py --steps 96 --warmup 3 --repeats 5 --no-cpu-baseline --views-per-pass $1 --frames-in-flight $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['frames_in_flight'], d['config']['views_per_pass'])" || exit 1
done
