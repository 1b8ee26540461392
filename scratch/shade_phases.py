"""Where rt_shade's cycles go (RT_STATS build): prefix table, shading loop, Phong pass, set-up of the next depth's rays.
Usage: python scratch/shade_phases.py [views]"""
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
L.rt_debug_shade_timing.argtypes = [C.c_void_p, C.c_size_t]
tot = np.zeros(5)
for level in range(0, 11):
    os.environ['RT_DEBUG_STOP_LEVEL'] = str(level)
    # the stop level ends the frame after rt_traverse of that depth: rt_shade of the depth before is the last one that ran
    for _ in range(2):
        dev.render_device(buf.data_ptr(), max_depth=10); torch.cuda.synchronize()
    if level == 0: continue
    ts = np.zeros((8192, 8), dtype=np.uint64)
    L.rt_debug_shade_timing(ts.ctypes.data, ts.size)
    ts = ts[ts[:, 0] > 0].astype(np.float64)
    d = np.diff(ts[:, :6], axis=1)
    tot += d.sum(axis=0)
    print('rt_shade at depth %2d | waves %d | wave cycles mean %.0f max %.0f | prefix %.0f%% shading %.0f%% stats+barrier %.0f%% phong %.0f%% setup %.0f%%'
          % (level - 1, len(ts), (ts[:, 5] - ts[:, 0]).mean(), (ts[:, 5] - ts[:, 0]).max(), *(100 * d.sum(axis=0) / d.sum())))
print('all depths but the last: prefix %.0f%% shading %.0f%% stats+barrier %.0f%% phong %.0f%% setup %.0f%%' % tuple(100 * tot / tot.sum()))
