# Random triangle soups as meshes: 1 to 400 triangles of very different sizes, duplicated triangles (ties inside one
# mesh), zero-area triangles, coplanar overlaps; two to five instances with random transforms.  Device against oracle.
import sys, os, tempfile, json, random
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import orc
from rust_raytracer_amd import synthetic
import test_gpu_parity as T
a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp()
bad, refused = [], 0
for seed in range(a, b):
    rng = random.Random(seed * 15485863 + 3)
    u = rng.uniform
    scene = json.loads(synthetic.random_scene(d, seed))
    n = rng.choice([1, 2, 6, 7, 40, 400])
    size = rng.choice([1e-3, 0.05, 0.3, 1.5, 5.0])
    lines, tris = [], []
    for k in range(n):
        c = (u(-1, 1), u(-1, 1), u(-1, 1))
        r = rng.random()
        if r < 0.1 and tris: tri = list(tris[rng.randrange(len(tris))])                 # an exact duplicate
        elif r < 0.2: p = [c[i] + u(-size, size) for i in range(3)]; tri = [p, p, [c[i] + u(-size, size) for i in range(3)]]   # zero area
        elif r < 0.3: tri = [[c[0] + u(-size, size), c[1] + u(-size, size), 0.25] for _ in range(3)]                         # all in one plane
        else: tri = [[c[i] + u(-size, size) for i in range(3)] for _ in range(3)]
        tris.append(tri)
    for tri in tris:
        for p in tri: lines.append('v %.6f %.6f %.6f' % tuple(p))
    for k in range(len(tris)):
        nn = (u(-1, 1), u(-1, 1), u(0.2, 1))
        for _ in range(3): lines.append('vn %.6f %.6f %.6f' % nn)
    for k in range(len(tris)):
        i = 3 * k + 1
        lines.append('f %d//%d %d//%d %d//%d' % (i, i, i + 1, i + 1, i + 2, i + 2))
    name = 'soup_%d.obj' % seed
    open(os.path.join(d, name), 'w').write('\n'.join(lines) + '\n')
    opaque = [m['name'] for m in scene['materials'] if isinstance(m['material_type'], str)]
    scene['meshes'].append({'name': 'soup', 'file_name': name, 'winding_order': rng.choice(['Clockwise', 'CounterClockwise']), 'material': rng.choice(opaque)})
    for _ in range(rng.randint(2, 5)):
        scene['mesh_instances'].append({'mesh': 'soup', 'position': synthetic._xyz(u(-3, 3), u(-1, 3), u(-9, -3)),
                                        'orientation_degrees': synthetic._xyz(u(0, 360), u(0, 360), u(0, 360)),
                                        'scale': synthetic._xyz(u(0.2, 2), u(0.2, 2), u(0.2, 2))})
    try:
        s = orc.OracleScene.from_text(json.dumps(scene), d)
    except Exception as e:
        refused += 1; os.remove(os.path.join(d, name)); continue
    got, want = T.render_both(s, max_depth=5)
    try:
        T.assert_parity(got, want, min_within_1=0.99)
    except AssertionError as e:
        bad.append(seed); print('seed', seed, 'FAILED', (n, size), str(e)[:200], flush=True)
    os.remove(os.path.join(d, name))
    if seed % 100 == 0: print('at seed', seed, flush=True)
print('seeds', a, b, 'refused by the BVH builder', refused, 'failures', bad)
