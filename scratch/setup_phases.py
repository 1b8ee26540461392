"""Where the set-up of a depth's rays spends a wave's cycles (RT_STATS build): fetching the chunk's rays, the analytic
primitives, the culling boxes of the instances, pushing / draining pairs and storing the key.
Usage: python scratch/setup_phases.py [views]"""
import sys, os, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from rust_raytracer_amd import meshes, native
native.LIB_PATH = os.path.abspath('scratch/librt_hip_stats.so')
views = int(sys.argv[1]) if len(sys.argv) > 1 else 1
md = meshes.ensure_model_dir()
host = native.HostScene.load('scenes/reflective_bunnies.json', md)
host.resize_screen(1920, 1080)
dev = native.DeviceScene(host.desc(), 0)
m, w, h = host.camera(); dev.set_views(np.tile(np.asarray(m, dtype=np.float32).reshape(1, 16), (views, 1)), w, h)
buf = torch.zeros(views * w * h * 3, dtype=torch.uint8, device='cuda')
L = native.lib()
L.rt_debug_setup_timing.argtypes = [C.c_void_p, C.c_size_t]
names = ['fetch', 'analytic', 'cull', 'push+drain+key']
tot = np.zeros(5)
for level in range(0, 11):
    os.environ['RT_DEBUG_STOP_LEVEL'] = str(level)
    for _ in range(2):
        dev.render_device(buf.data_ptr(), max_depth=10); torch.cuda.synchronize()
    ts = np.zeros((8192, 8), dtype=np.uint64)
    L.rt_debug_setup_timing(ts.ctypes.data, ts.size)
    ts = ts[ts[:, 4] > 0].astype(np.float64)
    chunks = ts[:, 4].sum()
    per = ts[:, :4].sum(axis=0) / chunks
    tot[:4] += ts[:, :4].sum(axis=0); tot[4] += chunks
    print('set-up of depth %2d | waves %d | chunks per wave %.1f | cycles per chunk %.0f: %s'
          % (level, len(ts), chunks / len(ts), per.sum(), ' '.join('%s %.0f' % (n, v) for n, v in zip(names, per))))
print('all depths: cycles per chunk %.0f: %s' % (tot[:4].sum() / tot[4], ' '.join('%s %.0f' % (n, v / tot[4]) for n, v in zip(names, tot[:4]))))
