import sys, os, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from rust_raytracer_amd import meshes, native
native.LIB_PATH = os.path.abspath('scratch/librt_hip_stats.so')
md = meshes.ensure_model_dir()
host = native.HostScene.load('scenes/reflective_bunnies.json', md)
host.resize_screen(1920, 1080)
dev = native.DeviceScene(host.desc(), 0)
m, w, h = host.camera(); dev.set_camera(m, w, h)
for _ in range(2): r = dev.render(max_depth=10)
print('trace ms', dev.last_trace_ms(), r['counters'][:5])
out = (C.c_ulonglong * (64 * 7))()
native.lib().rt_debug_stats.argtypes = [C.c_void_p, C.c_void_p]
native.lib().rt_debug_stats(dev._h, out)
a = np.array(out).reshape(64, 7)
print('level inner leaf tris setups rounds(wave) transitions iters(wave)')
for l in range(10): print(l, a[l].tolist())
print('sum', a.sum(0).tolist())
