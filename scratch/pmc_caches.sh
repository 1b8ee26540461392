#!/bin/bash
# On the GPU box: address-unit and cache counters per kernel, five headline frames one at a time -> gpurun_out/pmc_caches.txt
set -e
W=${1:-reflective_bunnies_1080p}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc1 /tmp/pc2
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/pc1 -- python3 /root/repo/scratch/frames.py $W 5 > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/pc2 -- python3 /root/repo/scratch/frames.py $W 5 > /dev/null 2>&1
{ python3 /root/repo/scratch/pmc_sq.py /tmp/pc1; python3 /root/repo/scratch/pmc_sq.py /tmp/pc2; } | grep -v "per wave" > /root/repo/gpurun_out/pmc_caches_$W.txt
