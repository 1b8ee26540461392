# Depth-map mode (Raytracer::depth_map) on random and spoiled scenes, device against oracle.
import sys, os, tempfile
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import orc
from rust_raytracer_amd import native, synthetic
a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp()
bad = []
for seed in range(a, b):
    text = synthetic.spoiled_scene(d, seed)[0] if seed % 2 else synthetic.random_scene(d, seed)
    s = orc.OracleScene.from_text(text, d)
    m, w, h = s.camera()
    dev = native.DeviceScene(s.desc(), 0)
    try:
        dev.set_camera(m, w, h)
        got = dev.render(depth_map=True)
    finally:
        dev.close()
    want = s.render_depth_map()
    diff = np.abs(got['rgb'].astype(int) - want['rgb'].astype(int)).max(axis=2)
    if not ((diff <= 1).mean() >= 0.999 and got['counters'][:5].tolist() == [w * h, w * h, 0, 0, 0]):
        bad.append(seed); print('seed', seed, 'FAILED within %.4f max %d' % ((diff <= 1).mean(), diff.max()), flush=True)
    if seed % 200 == 0: print('at seed', seed, flush=True)
print('seeds', a, b, 'failures', bad)
