import sys, time, os
sys.path.insert(0, '.')
import numpy as np
from oracle import orc
from rust_raytracer_amd import meshes, native, abi
md = meshes.ensure_model_dir()
s = orc.OracleScene.from_file('scenes/reflective_bunnies.json', md)
s.resize(1920, 1080)
m, w, h = s.camera()
dev = native.DeviceScene(s.desc(), 0)
dev.set_camera(m, w, h)
for i in range(3):
    t = time.time(); got = dev.render(aux=True); dt = time.time() - t
    print('aux render wall %.1f ms, kernel %.3f ms' % (dt * 1e3, dev.last_render_ms()), got['counters'][:5])
import ctypes as C
for i in range(5):
    t = time.time(); got2 = dev.render(aux=False); dt = time.time() - t
    ms = dev.last_render_ms()
    print('render wall %.1f ms, kernel %.3f ms -> %.1f Mrays/s' % (dt * 1e3, ms, got2['counters'][0] / ms / 1e3))
t = time.time(); want = s.render(10); dt = time.time() - t
print('oracle %.2f s  %.2f Mrays/s on %d threads' % (dt, want['counters'][0] / dt / 1e6, os.cpu_count()))
for k in ('flags', 'prim_id', 'instance_id'):
    print(k, 'mismatch', int((got[k] != want[k]).sum()))
print('counters', got['counters'][:5].tolist(), want['counters'][:5].tolist())
d = np.abs(got['rgb'].astype(int) - want['rgb'].astype(int)).max(axis=2)
print('rgb exact %.6f within1 %.6f max %d' % ((d == 0).mean(), (d <= 1).mean(), d.max()))
print('spilled', dev.debug_last_regrown())
