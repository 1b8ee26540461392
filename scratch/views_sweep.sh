#!/bin/bash
# On the GPU box: the headline bench line for several (views per pass, passes in flight, list capacity in rays per pixel and depth)
for cfg in "8 3 8" "8 3 4" "16 3 4" "16 4 4" "24 2 4" "12 4 4"; do
  set -- $cfg
  echo "== views-per-pass $1 frames-in-flight $2 rays per pixel and depth $3"
  RT_QUEUE_CAPACITY=$(( $3 * $1 * 2073600 )) timeout -k 10 200 python bench.py --steps 96 --warmup 3 --repeats 5 --no-cpu-baseline --views-per-pass $1 --frames-in-flight $2 2>gpurun_out/views_sweep.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['frames_in_flight'], d['config']['views_per_pass'])" || { tail -3 gpurun_out/views_sweep.err; }
done
