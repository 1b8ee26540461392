#!/bin/bash
# What profiles/r03_* holds, in parts (each fits one call on the GPU box): bash scratch/profile_round3.sh <part>   -> gpurun_out/r03/
#   1  kernel trace summaries of the default bench command (queued) and of one frame at a time
#   2  SQ / cache / memory-path counters of the headline frame, one frame at a time and 16 views per pass
#   3  HBM traffic per launch (FETCH_SIZE / WRITE_SIZE passes) for every workload, and queued for the headline
#   4  bench.py for every workload, the collective exercised at world size 1, per-kernel times
#   5  full-size parity, phase timings (RT_STATS build), list use
part=$1
OUT=/root/repo/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
H=reflective_bunnies_1080p
case $part in
1)
  rm -rf /tmp/ks1 /tmp/ks2
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -- python3 /root/repo/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || exit 1
  cp $(find /tmp/ks1 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks2 -- python3 /root/repo/bench.py --no-cpu-baseline --frames-in-flight 1 --views-per-pass 1 > $OUT/bench_one_frame_at_a_time.json 2> $OUT/bench_one_frame_at_a_time.err || exit 1
  cp $(find /tmp/ks2 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_one_frame_at_a_time.csv
  head -8 $OUT/kernel_stats.csv
  ;;
2)
  pass() { # <dir> <views> <counters...>
    d=$1; v=$2; shift 2
    rm -rf $d
    VIEWS=$v timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $d -- python3 /root/repo/scratch/frames.py $H 5 > /dev/null 2>&1 || return 1
  }
  for v in 1 16; do
    tag=$([ $v = 1 ] && echo one_frame_at_a_time || echo 16_views_per_pass)
    pass /tmp/sq1 $v SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES && python3 /root/repo/scratch/pmc_sq.py /tmp/sq1 > $OUT/pmc_instruction_mix_$tag.txt || exit 1
    pass /tmp/sq2 $v SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES && python3 /root/repo/scratch/pmc_sq.py /tmp/sq2 > $OUT/pmc_wave_cycles_$tag.txt || exit 1
    pass /tmp/sq3 $v GRBM_GUI_ACTIVE GRBM_COUNT && python3 /root/repo/scratch/pmc_sq.py /tmp/sq3 > $OUT/pmc_grbm_$tag.txt || exit 1
    pass /tmp/pc1 $v TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE || exit 1
    pass /tmp/pc2 $v TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE || exit 1
    { python3 /root/repo/scratch/pmc_sq.py /tmp/pc1; python3 /root/repo/scratch/pmc_sq.py /tmp/pc2; } | grep -v "per wave" > $OUT/pmc_caches_$tag.txt
    echo "counters at $v view(s) per pass done"
  done
  ;;
3)
  cd /root/repo
  bash scratch/pmc_traffic.sh reflective_bunnies_1080p reflective_bunnies_1080p_sphere_standin cornell_box_720p mirror_sphere_1080p_depth16 infinity_santa_1080p_depth16 instance_grid_2p6m_4k instance_grid_10m_4k || exit 1
  bash scratch/pmc_traffic_queued.sh 16 reflective_bunnies_1080p || exit 1
  cp gpurun_out/traffic_*.json $OUT/
  ;;
4)
  cd /root/repo
  for w in reflective_bunnies_1080p reflective_bunnies_1080p_sphere_standin cornell_box_720p mirror_sphere_1080p_depth16 infinity_santa_1080p_depth16 instance_grid_2p6m_4k instance_grid_10m_4k; do
    timeout -k 10 500 python3 bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err || { echo "$w FAILED"; tail -3 $OUT/bench_$w.err; exit 1; }
    echo "$w done"
  done
  timeout -k 10 400 python3 bench.py --exercise-collective --no-cpu-baseline > $OUT/bench_exercise_collective.json 2> $OUT/bench_exercise_collective.err || { tail -5 $OUT/bench_exercise_collective.err; exit 1; }
  echo "collective done"
  ;;
5)
  cd /root/repo
  timeout -k 10 700 python3 scratch/full_size_parity.py > $OUT/full_size_parity.txt 2>&1 || { tail -5 $OUT/full_size_parity.txt; exit 1; }
  echo "parity done"
  timeout -k 10 600 python3 scratch/kernel_times.py reflective_bunnies_1080p cornell_box_720p mirror_sphere_1080p_depth16 infinity_santa_1080p_depth16 instance_grid_2p6m_4k instance_grid_10m_4k 2>/dev/null > $OUT/kernel_times.txt
  VIEWS=16 timeout -k 10 300 python3 scratch/kernel_times.py reflective_bunnies_1080p 2>/dev/null > $OUT/kernel_times_16_views.txt
  { timeout -k 10 200 python3 scratch/shade_phases.py 1; timeout -k 10 200 python3 scratch/shade_phases.py 16; } > $OUT/shade_phases.txt 2>/dev/null
  { timeout -k 10 200 python3 scratch/setup_phases.py 1; timeout -k 10 200 python3 scratch/setup_phases.py 16; } > $OUT/setup_phases.txt 2>/dev/null
  timeout -k 10 200 python3 scratch/list_use.py 2> $OUT/list_use_per_depth.txt > $OUT/list_use.txt
  echo "phases done"
  ;;
esac
