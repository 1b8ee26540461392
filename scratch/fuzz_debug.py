import sys, os, tempfile
sys.path.insert(0, '/root/repo')
import numpy as np
from rust_raytracer_amd import synthetic, native
from oracle import orc
seed = int(sys.argv[1]); depth = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = tempfile.mkdtemp()
s = orc.OracleScene.from_text(synthetic.random_scene(d, seed), d)
m, w, h = s.camera()
dev = native.DeviceScene(s.desc(), 0); dev.set_camera(m, w, h)
for dep in range(1, depth + 1):
    got = dev.render(max_depth=dep, aux=True); want = s.render(max_depth=dep)
    diff = np.abs(got['rgb'].astype(int) - want['rgb'].astype(int)).max(axis=2)
    bad = np.argwhere(diff > 1)
    print('depth', dep, 'bad pixels', len(bad), 'counters equal', got['counters'][:5].tolist() == want['counters'][:5].tolist())
    if len(bad):
        for (y, x) in bad[:6]:
            c = s.trace_pixel(float(x), float(h - 1 - y), dep)
            print('   pixel', x, y, 'got', got['rgb'][y, x].tolist(), 'want', want['rgb'][y, x].tolist(), 'oracle colour', c, 'prim', hex(int(want['prim_id'][y, x])), 'flags', int(want['flags'][y, x]))
        break
