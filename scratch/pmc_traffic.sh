#!/bin/bash
# HBM-side bytes per launch and kernel for bench.py workloads (separate FETCH_SIZE / WRITE_SIZE passes, as the counters
# do not fit one pass): bash scratch/pmc_traffic.sh <workload> ...   -> gpurun_out/traffic_<workload>.json
set -e
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  rm -rf /tmp/pf_$w /tmp/pw_$w
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$w -- python3 /root/repo/scratch/frames.py $w > /dev/null 2>&1
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw_$w -- python3 /root/repo/scratch/frames.py $w > /dev/null 2>&1
  python3 /root/repo/scratch/pmc_summary.py /tmp/pf_$w /tmp/pw_$w /root/repo/gpurun_out/traffic_$w.json "$w" > /dev/null
  echo "$w done"
done
