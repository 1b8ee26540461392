# Random scenes pushed to extremes: orthographic cameras, extreme fields of view, homogeneous light positions with
# w other than 0 and 1, instances far away / huge / tiny, huge and tiny spheres, far planes.  Device against oracle.
import sys, os, tempfile, json, random
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import orc
from rust_raytracer_amd import synthetic
import test_gpu_parity as T
a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp()
bad = []
for seed in range(a, b):
    rng = random.Random(seed * 7919 + 13)
    scene = json.loads(synthetic.random_scene(d, seed))
    what = []
    cam = scene['camera']
    if rng.random() < 0.3: cam['use_perspective'] = False; cam['ortho_world_height'] = rng.choice([0.5, 10, 200]); what.append('ortho')
    if rng.random() < 0.2: cam['y_fov_degrees'] = rng.choice([1.0, 5.0, 150.0, 178.0]); what.append('fov')
    if rng.random() < 0.2: cam['z_near'] = rng.choice([1e-4, 5.0]); cam['z_far'] = rng.choice([6.0, 1e6]); what.append('near/far')
    for L in scene['lights']:
        if rng.random() < 0.3: L['position']['w'] = rng.choice([2.0, -1.0, 0.5, 1e-3]); what.append('light w')
    for inst in scene['mesh_instances']:
        r = rng.random()
        if r < 0.15:
            k = rng.choice([1e3, 1e5]); inst['position'] = {c: inst['position'][c] * k for c in 'xyz'}; what.append('far instance')
        elif r < 0.3:
            k = rng.choice([1e-3, 1e3]); inst['scale'] = {c: inst['scale'][c] * k for c in 'xyz'}; what.append('instance scale %g' % k)
    for sp in scene['spheres']:
        if rng.random() < 0.2: sp['radius'] = rng.choice([1e-4, 1e4]); what.append('sphere radius')
    for pl in scene['planes']:
        if rng.random() < 0.2: pl['distance'] = rng.choice([-1e6, 1e6, 0.0]); what.append('plane distance')
    s = orc.OracleScene.from_text(json.dumps(scene), d)
    got, want = T.render_both(s, max_depth=5)
    try:
        T.assert_parity(got, want, min_within_1=0.98)
    except AssertionError as e:
        bad.append(seed); print('seed', seed, 'FAILED', what, str(e)[:200], flush=True)
    if seed % 100 == 0: print('at seed', seed, flush=True)
print('seeds', a, b, 'failures', bad)
