# Scenes with many analytic primitives (more than rt_setup stages in LDS) and many lights.  Device against oracle.
import sys, os, tempfile, json, random
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import orc
from rust_raytracer_amd import synthetic
import test_gpu_parity as T
a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp()
bad = []
for seed in range(a, b):
    rng = random.Random(seed * 104729 + 7)
    scene = json.loads(synthetic.random_scene(d, seed, 80, 48))
    u = rng.uniform
    mats = [m['name'] for m in scene['materials']]
    n_s, n_t, n_p, n_l = rng.choice([0, 50, 900, 1500]), rng.choice([0, 30, 250]), rng.choice([0, 40, 300]), rng.choice([0, 3, 12, 40])
    for _ in range(n_s):
        scene['spheres'].append({'center': synthetic._xyz(u(-8, 8), u(-2, 6), u(-16, -3)), 'radius': u(0.02, 0.3), 'material': rng.choice(mats)})
    for _ in range(n_t):
        c = (u(-6, 6), u(-2, 5), u(-14, -4))
        vs = [synthetic._xyz(c[0] + u(-.5, .5), c[1] + u(-.5, .5), c[2] + u(-.5, .5)) for _ in range(3)]
        n = synthetic._xyz(u(-1, 1), u(-1, 1), u(0.1, 1))
        opaque = [m['name'] for m in scene['materials'] if isinstance(m['material_type'], str)]
        scene['triangles'].append({'vertices': vs, 'normals': [n, n, n], 'material': rng.choice(opaque)})
    for _ in range(n_p):
        scene['planes'].append({'normal': synthetic._xyz(u(-1, 1), u(-1, 1), u(-1, 1)), 'distance': u(-60, -20), 'material': rng.choice(mats)})
    for _ in range(n_l):
        scene['lights'].append({'position': {'x': u(-8, 8), 'y': u(1, 9), 'z': u(-12, 4), 'w': rng.choice([0, 1])},
                                'color': {'ambient': synthetic._rgb(0.01, 0.01, 0.01), 'diffuse': synthetic._rgb(u(0, .1), u(0, .1), u(0, .1)), 'specular': synthetic._rgb(0.02, 0.02, 0.02)}})
    s = orc.OracleScene.from_text(json.dumps(scene), d)
    got, want = T.render_both(s, max_depth=4)
    try:
        T.assert_parity(got, want, min_within_1=0.99)
    except AssertionError as e:
        bad.append(seed); print('seed', seed, 'FAILED', (n_s, n_t, n_p, n_l), str(e)[:200], flush=True)
    if seed % 20 == 0: print('at seed', seed, flush=True)
print('seeds', a, b, 'failures', bad)
