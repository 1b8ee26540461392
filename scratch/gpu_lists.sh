#!/bin/bash
# the list-sizing tests, then the suite's overflow tests, then a bench line
mkdir -p gpurun_out/lists
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -v --timeout 300 --timeout-method thread -m gpu \
  -k "sized_by or five_4k or overflow or orbit_batch" > gpurun_out/lists/tests.log 2>&1
rc=$?
tail -25 gpurun_out/lists/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 48 --warmup 16 > gpurun_out/lists/bench48.json 2> gpurun_out/lists/bench48.err
rc=$?
tail -c 1500 gpurun_out/lists/bench48.json
exit $rc
