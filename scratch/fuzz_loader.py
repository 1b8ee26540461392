# The library's scene loader against the oracle's on random and spoiled scenes (CPU only): same arrays, bit for bit.
import sys, os, tempfile
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from oracle import orc
from rust_raytracer_amd import abi, native, synthetic
import test_host_loader as T
a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp()
bad = []
for seed in range(a, b):
    text = synthetic.spoiled_scene(d, seed)[0] if seed % 2 else synthetic.random_scene(d, seed)
    path = os.path.join(d, 's.json'); open(path, 'w').write(text)
    try:
        mine = native.HostScene.load(path, d)
    except native.RtError as e:
        mine = None
    try:
        ref = orc.OracleScene.from_file(path, d)
    except Exception as e:
        ref = None
    if (mine is None) != (ref is None):
        bad.append(seed); print('seed', seed, 'one loader refused the scene', mine is None, ref is None, flush=True); continue
    if mine is None: continue
    x, y = abi.desc_arrays(mine.desc()), abi.desc_arrays(ref.desc())
    same = x['background'] == y['background'] and all(x[k] == y[k] for k in T.TYPES)        # raw bytes
    ma, wa, ha = mine.camera(); mb, wb, hb = ref.camera()
    same = same and (wa, ha) == (wb, hb) and np.array_equal(np.asarray(ma, dtype=np.float32).view(np.uint32), np.asarray(mb, dtype=np.float32).view(np.uint32))
    if not same:
        bad.append(seed); print('seed', seed, 'arrays differ:', [k for k in T.TYPES if x[k] != y[k]], flush=True)
    mine.close(); ref.close()
print('seeds', a, b, 'failures', bad)
