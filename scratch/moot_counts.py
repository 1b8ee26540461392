"""How many shadow rays per depth are not traced because their verdict cannot change the Phong sum (RT_STATS build), and how
full the chunks of 64 are that remain.  Usage: python scratch/moot_counts.py"""
import sys, os, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from rust_raytracer_amd import meshes, native
native.LIB_PATH = os.path.abspath('scratch/librt_hip_stats.so')
md = meshes.ensure_model_dir()
host = native.HostScene.load('scenes/reflective_bunnies.json', md)
host.resize_screen(1920, 1080)
dev = native.DeviceScene(host.desc(), 0)
m, w, h = host.camera(); dev.set_camera(m, w, h)
buf = torch.zeros(w * h * 3, dtype=torch.uint8, device='cuda')
L = native.lib()
L.rt_debug_moot_counts.argtypes = [C.c_void_p, C.c_size_t]
dev.render_device(buf.data_ptr(), max_depth=10); torch.cuda.synchronize()
c = np.zeros(2 * 65, dtype=np.uint64)
L.rt_debug_moot_counts(c.ctypes.data, c.size)
c = c.reshape(-1, 2).astype(np.int64)
for d, (a, mo) in enumerate(c[:64]):
    if a:
        print('depth %2d: %9d shadow rays, %9d not traced (%.1f %%)' % (d, a, mo, 100.0 * mo / a))
print('all: %d of %d (%.1f %%)' % (c[:64, 1].sum(), c[:64, 0].sum(), 100.0 * c[:64, 1].sum() / c[:64, 0].sum()))
