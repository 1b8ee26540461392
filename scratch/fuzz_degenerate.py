# synthetic.spoiled_scene over seeds [a, b): random scenes with duplicates (ties in t), zero scales, zero radius, zero
# normals, collapsed triangles, a light at the camera.  Device against oracle.
import sys, os, tempfile
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import orc
from rust_raytracer_amd import synthetic
import test_gpu_parity as T
a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp()
bad = []
for seed in range(a, b):
    text, what = synthetic.spoiled_scene(d, seed)
    s = orc.OracleScene.from_text(text, d)
    got, want = T.render_both(s, max_depth=5)
    try:
        T.assert_parity(got, want, min_within_1=0.99)
    except AssertionError as e:
        bad.append(seed); print('seed', seed, 'FAILED', what, str(e)[:300], flush=True)
    if seed % 100 == 0: print('at seed', seed, flush=True)
print('seeds', a, b, 'failures', bad)
