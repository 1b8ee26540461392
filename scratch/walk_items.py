"""Walk items ((ray, instance) pairs that passed the root-box test) per depth against the rays of that depth (RT_STATS build)."""
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
r = dev.render(max_depth=10)
L = native.lib()
L.rt_debug_walk_items.argtypes = [C.c_void_p, C.c_size_t]
it = np.zeros(64, dtype=np.uint32)
L.rt_debug_walk_items(it.ctypes.data, it.size)
c = r['counters']
print('walk items per depth', it[:11].tolist(), 'total', int(it[:11].sum()))
print('rays', int(c[0]), 'node visits', int(c[native.RT_CTR_WALK_NODE_VISITS]) if hasattr(native, 'RT_CTR_WALK_NODE_VISITS') else c.tolist())
