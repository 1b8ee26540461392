# Walks of more than 300 node visits in a frame at drag camera k (RT_STATS build): python scratch/long_walks.py <k>
import sys, os, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np
from rust_raytracer_amd import meshes, native
native.LIB_PATH = os.path.abspath('/root/repo/scratch/librt_hip_stats.so')
k = int(sys.argv[1])
host = native.HostScene.load('/root/repo/scenes/reflective_bunnies.json', meshes.ensure_model_dir())
host.resize_screen(1920, 1080)
if k:
    step = 12.0 * ((k + 1) // 2)
    host.turn_camera(0.0, 0.0, step if k % 2 else -step, 0.0)
m, w, h = host.camera()
dev = native.DeviceScene(host.desc(), 0)
dev.set_camera(m, w, h)
r = dev.render(max_depth=10)
L = native.lib()
L.rt_debug_long_walks.argtypes = [C.c_void_p, C.c_size_t]
buf = np.zeros(1 + 64 * 24, dtype=np.uint32)
L.rt_debug_long_walks(buf.ctypes.data, buf.size)
print('camera', k, 'long walks', int(buf[0]))
for i in range(min(int(buf[0]), 12)):
    o = buf[1 + 24 * i: 1 + 24 * (i + 1)]
    f = o[8:].view(np.float32)
    print(' level %d item %d ray %d instance %d | local origin %s root tmin %g | dir %s limit %g | 1/dn %g thr %g prune_abs %g | flags %08x'
          % (o[0], o[1], o[2], o[3], f[0:3], f[3], f[4:7], f[7], f[8], f[9], f[10], o[8 + 13]))
