#!/bin/bash
# On the GPU box: what the vector memory path of a compute unit does during each kernel (address unit, L1, data return), five
# frames one at a time -> gpurun_out/<tag>/pmc_mempath_<workload>.txt          bash scratch/pmc_mempath.sh <tag> [workload]
set -e
TAG=$1
W=${2:-reflective_bunnies_1080p}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mp1 /tmp/mp2 /tmp/mp3
timeout -s KILL 120 rocprofv3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/mp1 -- python3 /root/repo/scratch/frames.py $W 5 > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TD_LOAD_WAVEFRONT_sum TD_COALESCABLE_WAVEFRONT_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/mp2 -- python3 /root/repo/scratch/frames.py $W 5 > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/mp3 -- python3 /root/repo/scratch/frames.py $W 5 > /dev/null 2>&1
{ for d in /tmp/mp1 /tmp/mp2 /tmp/mp3; do python3 /root/repo/scratch/pmc_sq.py $d || echo "pass $d gave nothing"; done; } | grep -v "per wave" > $OUT/pmc_mempath_$W.txt
