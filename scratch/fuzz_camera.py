# Degenerate cameras and frame sizes: field of view 0 / 180 / 360 degrees, z_near == z_far, z_near 0, frames of 1x1, 1xN,
# Nx1 and odd sizes - NaN and infinite primary rays included.  Device against oracle.
import sys, os, tempfile, json, random
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import orc
from rust_raytracer_amd import synthetic
import test_gpu_parity as T
a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp()
bad = []
for seed in range(a, b):
    rng = random.Random(seed * 32452843 + 5)
    w, h = rng.choice([(1, 1), (1, 37), (53, 1), (2, 2), (7, 5), (33, 31), (65, 9), (96, 64)])
    scene = json.loads(synthetic.spoiled_scene(d, seed, w, h)[0] if seed % 3 == 0 else synthetic.random_scene(d, seed, w, h))
    cam = scene['camera']
    what = [(w, h)]
    r = rng.random()
    if r < 0.15: cam['y_fov_degrees'] = rng.choice([0.0, 180.0, 360.0, -60.0]); what.append('fov %g' % cam['y_fov_degrees'])
    elif r < 0.3: cam['z_far'] = cam['z_near']; what.append('near == far')
    elif r < 0.4: cam['z_near'] = 0.0; what.append('near 0')
    elif r < 0.5: cam['use_perspective'] = False; cam['ortho_world_height'] = rng.choice([0.0, -5.0, 1e-6]); what.append('ortho height %g' % cam['ortho_world_height'])
    try:
        s = orc.OracleScene.from_text(json.dumps(scene), d)
    except Exception as e:
        print('seed', seed, what, 'refused:', str(e)[:100], flush=True); continue
    got, want = T.render_both(s, max_depth=5)
    try:
        T.assert_parity(got, want, min_within_1=0.98 if w * h > 400 else 0.9)
    except AssertionError as e:
        bad.append(seed); print('seed', seed, 'FAILED', what, str(e)[:200], flush=True)
    if seed % 200 == 0: print('at seed', seed, flush=True)
print('seeds', a, b, 'failures', bad)
