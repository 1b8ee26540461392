# Everything at once (synthetic.mixed_scene): a spoiled scene that also gets extreme parameters, a triangle soup as a
# further mesh, a random frame size and depth.  Device against oracle.
import sys, os, tempfile, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import orc
from rust_raytracer_amd import synthetic
import test_gpu_parity as T
a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp()
bad, refused = [], 0
for seed in range(a, b):
    text, depth, what = synthetic.mixed_scene(d, seed)
    w, h = json.loads(text)['camera']['screen_dimensions']
    try:
        s = orc.OracleScene.from_text(text, d)
    except Exception:
        refused += 1; continue
    got, want = T.render_both(s, max_depth=depth)
    try:
        T.assert_parity(got, want, min_within_1=0.98 if w * h > 2000 else 0.9)
    except AssertionError as e:
        bad.append(seed); print('seed', seed, 'FAILED', what, str(e)[:200], flush=True)
    if seed % 200 == 0: print('at seed', seed, flush=True)
print('seeds', a, b, 'refused', refused, 'failures', bad)
