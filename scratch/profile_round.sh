#!/bin/bash
# Everything profiles/ holds for a round, in one go on the GPU box: bash scratch/profile_round.sh <tag>   (output: gpurun_out/<tag>/)
set -e
TAG=$1
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the default bench run under the kernel trace (three passes in flight: launches overlap), and one frame at a time
rm -rf /tmp/ks1 /tmp/ks2
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -- python3 /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/ks1 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks2 -- python3 /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --frames-in-flight 1 --views-per-pass 1 > $OUT/bench_one_frame_at_a_time.json 2> /dev/null
cp $(find /tmp/ks2 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_one_frame_at_a_time.csv
echo "kernel stats done"
# 2. SQ counters of the headline frame, one frame at a time (8 SQ counters per pass at most; GRBM separately)
rm -rf /tmp/sq1 /tmp/sq2 /tmp/sq3
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/sq1 -- python3 /root/repo/scratch/frames.py reflective_bunnies_1080p 5 > /dev/null 2>&1
python3 /root/repo/scratch/pmc_sq.py /tmp/sq1 > $OUT/pmc_instruction_mix.txt
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES --output-format csv -d /tmp/sq2 -- python3 /root/repo/scratch/frames.py reflective_bunnies_1080p 5 > /dev/null 2>&1
python3 /root/repo/scratch/pmc_sq.py /tmp/sq2 > $OUT/pmc_wave_cycles.txt
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d /tmp/sq3 -- python3 /root/repo/scratch/frames.py reflective_bunnies_1080p 5 > /dev/null 2>&1
python3 /root/repo/scratch/pmc_sq.py /tmp/sq3 > $OUT/pmc_grbm.txt
echo "sq counters done"
# 3. per-kernel, per-depth durations of a lone 1/8 shard and of the whole frame
bash /root/repo/scratch/trace_both.sh $TAG/r
# 4. per-kernel ms per frame for every workload, one view and eight views per pass
timeout -k 10 600 python3 /root/repo/scratch/kernel_times.py reflective_bunnies_1080p cornell_box_720p mirror_sphere_1080p_depth16 infinity_santa_1080p_depth16 instance_grid_2p6m_4k instance_grid_10m_4k 2>/dev/null > $OUT/kernel_times.txt
VIEWS=8 timeout -k 10 300 python3 /root/repo/scratch/kernel_times.py reflective_bunnies_1080p 2>/dev/null > $OUT/kernel_times_8_views.txt
echo "kernel times done"
