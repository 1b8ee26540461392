#!/bin/bash
# HBM-side bytes per launch and kernel for passes of several frames (the mode bench.py times `value` in):
# bash scratch/pmc_traffic_queued.sh <views> <workload> ...   -> gpurun_out/traffic_queued_<workload>.json
set -e
V=$1; shift
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  rm -rf /tmp/qf_$w /tmp/qw_$w
  VIEWS=$V timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/qf_$w -- python3 /root/repo/scratch/frames.py $w > /dev/null 2>&1
  VIEWS=$V timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/qw_$w -- python3 /root/repo/scratch/frames.py $w > /dev/null 2>&1
  python3 /root/repo/scratch/pmc_summary.py /tmp/qf_$w /tmp/qw_$w /root/repo/gpurun_out/traffic_queued_$w.json "$w" $V > /dev/null
  echo "$w done"
done
