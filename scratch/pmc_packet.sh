#!/bin/bash
# On the GPU box: the BVH walk of the first depths as a packet walk against the lane-by-lane walk: per-depth launch times of one
# 1080p bunnies frame (rocprofv3 kernel trace) and the walk's vector / scalar memory instructions and address-unit time (PMC).
# bash scratch/pmc_packet.sh <tag>   -> gpurun_out/<tag>/packet_walk.txt
TAG=$1
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
{
for lv in 0 2; do
  echo "== RT_WALK_PACKET_LEVELS=$lv: per-launch durations of one frame (us)"
  rm -rf /tmp/pk$lv
  RT_WALK_PACKET_LEVELS=$lv timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/pk$lv -- python3 /root/repo/scratch/one_camera.py 0 > /dev/null 2>&1
  python3 /root/repo/scratch/trace_summary.py /tmp/pk$lv | sed -n 1,4p
  echo "== RT_WALK_PACKET_LEVELS=$lv: counters over 5 frames"
  rm -rf /tmp/pc$lv
  RT_WALK_PACKET_LEVELS=$lv timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/pc$lv -- python3 /root/repo/scratch/frames.py reflective_bunnies_1080p 5 > /dev/null 2>&1
  python3 /root/repo/scratch/pmc_sq.py /tmp/pc$lv | grep -v "per wave" | grep "rt_traverse"
done
} > $OUT/packet_walk.txt 2>&1
