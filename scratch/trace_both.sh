#!/bin/bash
# kernel-by-kernel durations of the last frame: a 1/8 shard rendered alone and the whole 1080p frame
# usage (on the GPU box): bash scratch/trace_both.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
for n in 8 1; do
  rm -rf /tmp/p$n
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/p$n -- python3 /root/repo/scratch/shard_kernels.py $n > /dev/null 2>&1
  python3 /root/repo/scratch/trace_summary.py /tmp/p$n > /root/repo/gpurun_out/$1_trace_shard$n.txt
done
