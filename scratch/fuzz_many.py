# A long run of the random-scene fuzz of tests/test_gpu_parity.py::test_random_scenes: seeds [a, b).
import sys, os, tempfile
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from oracle import orc
from rust_raytracer_amd import synthetic
import test_gpu_parity as T
a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp()
bad = []
for seed in range(a, b):
    s = orc.OracleScene.from_text(synthetic.random_scene(d, seed), d)
    got, want = T.render_both(s, max_depth=6)
    try:
        T.assert_parity(got, want, min_within_1=0.995)
    except AssertionError as e:
        bad.append(seed)
        print('seed', seed, 'FAILED', str(e)[:300], flush=True)
    if seed % 100 == 0: print('at seed', seed, flush=True)
print('seeds', a, b, 'failures', bad)
