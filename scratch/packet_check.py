# Packet walk against the oracle on an instance grid: python scratch/packet_check.py [levels] (library variant via native.LIB_PATH env RT_LIB)
import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
os.environ['RT_WALK_PACKET_LEVELS'] = sys.argv[1] if len(sys.argv) > 1 else '2'
from oracle import orc
from rust_raytracer_amd import native, synthetic
if os.environ.get('RT_LIB'): native.LIB_PATH = os.path.abspath(os.environ['RT_LIB'])
d = '/tmp/packet_check'; os.makedirs(d, exist_ok=True)
g = orc.OracleScene.from_text(synthetic.instance_grid_scene(d, 4, 4, 3, 320, 180), d)
m, w, h = g.camera()
dev = native.DeviceScene(g.desc(), 0)
dev.set_camera(m, w, h)
got = dev.render(max_depth=10, aux=True)
want = g.render(max_depth=10)
for k in ('flags', 'prim_id', 'instance_id'):
    print(k, 'differ on', int((got[k] != want[k]).sum()))
bad = np.argwhere(got['flags'] != want['flags'])
print('first bad pixels', bad[:5].tolist(), [(int(got['flags'][y, x]), int(want['flags'][y, x])) for y, x in bad[:5]])
print('counters', got['counters'][:5].tolist(), want['counters'][:5].tolist())
