"""Per block of rt_tail: cycles of its first step (shading of depth `first`), of every deeper depth's walk and shading (RT_STATS build)."""
import sys, os, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from rust_raytracer_amd import meshes, native
native.LIB_PATH = os.path.abspath('scratch/librt_hip_stats.so')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
md = meshes.ensure_model_dir()
host = native.HostScene.load('scenes/reflective_bunnies.json', md)
host.resize_screen(1920, 1080)
dev = native.DeviceScene(host.desc(), 0)
m, w, h = host.camera(); dev.set_camera(m, w, h)
buf = torch.zeros(dev.shard_bytes(n), dtype=torch.uint8, device='cuda')
for _ in range(3):
    dev.render_device(buf.data_ptr(), shard_index=0, shard_count=n); torch.cuda.synchronize()
print('frame ms', dev.last_render_ms())
L = native.lib()
L.rt_debug_tail_timing.argtypes = [C.c_void_p, C.c_size_t]
tt = np.zeros((2048, 40), dtype=np.uint64)
L.rt_debug_tail_timing(tt.ctypes.data, tt.size)
tt = tt[tt[:, 0] > 0].astype(np.float64)
print('blocks', len(tt))
tot = tt[:, 2] - tt[:, 0]
print('whole kernel per block: mean %.0f max %.0f min %.0f cycles (%.1f us max)' % (tot.mean(), tot.max(), tot.min(), tot.max() / 2400))
a = tt[:, 1] - tt[:, 0]
print('first step (shade of depth first + setup): mean %.0f max %.0f' % (a.mean(), a.max()))
for level in range(4, 11):
    walk = tt[:, 3 * level + 1] - tt[:, 3 * level]
    nxt = tt[:, 3 * (level + 1)] if level < 10 else tt[:, 2]
    shade = nxt - tt[:, 3 * level + 1]
    w = tt[:, 3 * level + 2].astype(np.uint64)
    items, rays, hits = (w >> np.uint64(40)), (w >> np.uint64(20)) & np.uint64(0xFFFFF), w & np.uint64(0xFFFFF)
    print('depth %2d: walk mean %.0f max %.0f | shade+phong+setup mean %.0f max %.0f | items mean %.0f max %d, rays mean %.0f max %d, hits mean %.0f max %d'
          % (level, walk.mean(), walk.max(), shade.mean(), shade.max(), items.mean(), items.max(), rays.mean(), rays.max(), hits.mean(), hits.max()))
