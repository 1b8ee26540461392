# Instance grids of random size (25 to 81 instances: the range tree over instances is in use) seen from random
# poses, device against oracle.
import sys, os, tempfile, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from oracle import orc
from rust_raytracer_amd import synthetic
import test_gpu_parity as T
a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp()
bad = []
for seed in range(a, b):
    rng = np.random.default_rng(seed)
    grid = int(rng.integers(5, 10)); n_meshes = int(rng.integers(1, 7)); level = int(rng.integers(1, 4))
    scene = json.loads(synthetic.instance_grid_scene(d, grid, n_meshes, level, int(rng.integers(80, 260)), int(rng.integers(60, 160)),
                                                     spacing=float(rng.uniform(1.2, 3.0))))
    cam = scene['camera']
    cam['position'] = {'x': float(rng.uniform(-6, 6)), 'y': float(rng.uniform(0.5, 6)), 'z': float(rng.uniform(-4, 4))}
    cam['orientation_degrees'] = {'x': float(rng.uniform(-40, 5)), 'y': float(rng.uniform(-40, 40)), 'z': float(rng.uniform(-10, 10))}
    s = orc.OracleScene.from_text(json.dumps(scene), d)
    got, want = T.render_both(s, max_depth=int(rng.integers(2, 9)))
    try:
        T.assert_parity(got, want, min_within_1=0.995)
    except AssertionError as e:
        bad.append(seed); print('seed', seed, 'FAILED', grid, n_meshes, level, str(e)[:300], flush=True)
    if seed % 100 == 0: print('at seed', seed, flush=True)
print('seeds', a, b, 'failures', bad)
