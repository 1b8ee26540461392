#!/bin/bash
mkdir -p gpurun_out/v256
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -m gpu -k "hundred_views or five_4k or api_errors or orbit or views" > gpurun_out/v256/tests.log 2>&1
rc=$?
tail -5 gpurun_out/v256/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/v256/bench.json 2> gpurun_out/v256/bench.err || { tail -5 gpurun_out/v256/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/v256/bench.json')); print(round(d['value']), d['ms_per_step'], d['steps'], d['config']['views_per_pass'])"
timeout -k 10 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --rehearse-on-one-gpu --no-cpu-baseline > gpurun_out/v256/out8.json 2> gpurun_out/v256/err8.txt
rc=$?
python -c "
import json; d=json.load(open('gpurun_out/v256/out8.json')); print('8 ranks rehearsed:', round(d['value']), d['ms_per_step'], d['steps'], d['config']['views_per_pass'], d['config']['cameras'][:40])" || { grep -v "socket.cpp\|amdgpu.ids\|Gloo" gpurun_out/v256/err8.txt | tail -8; exit 1; }
exit $rc
