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
L = native.lib()
L.rt_debug_items.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
for level in (1, 5):
    os.environ['RT_DEBUG_STOP_LEVEL'] = str(level)
    dev.render(max_depth=10)
    geo = (C.c_uint32 * 3)()
    L.rt_debug_items(dev._h, None, None, geo)
    n_seg, seg, ovf = geo
    items = np.zeros(((n_seg - 1) * seg + ovf, 2), dtype=np.uint32)
    counts = np.zeros(n_seg, dtype=np.uint32)
    L.rt_debug_items(dev._h, items.ctypes.data, counts.ctypes.data, geo)
    steps = np.concatenate([items[s * seg: s * seg + min(counts[s], seg if s < n_seg - 1 else ovf), 0] for s in range(n_seg)])
    print('level', level, 'items', len(steps), 'total steps', int(steps.sum()), 'mean %.1f' % steps.mean(), 'p50', int(np.percentile(steps, 50)), 'p90', int(np.percentile(steps, 90)), 'p99', int(np.percentile(steps, 99)), 'p99.9', int(np.percentile(steps, 99.9)), 'max', int(steps.max()), 'counts max/mean %.0f/%.0f' % (counts[:-1].max(), counts[:-1].mean()), 'ovf', counts[-1])
    tm = np.zeros((8192, 10), dtype=np.uint64)
    L.rt_debug_wave_timing.argtypes = [C.c_void_p, C.c_size_t]
    L.rt_debug_wave_timing(tm.ctypes.data, tm.size)
    tm = tm[tm[:, 1] > 0].astype(np.float64)
    names = ['cycles', 'rounds', 'refill_cyc', 'refills', 'leafround_cyc', 'leafrounds', 'leafload_cyc', 'leafmath_cyc', 'lanes_sum', 'leaf_lanes']
    print('   waves', len(tm), ' '.join('%s mean %.0f max %.0f |' % (n, tm[:, k].mean(), tm[:, k].max()) for k, n in enumerate(names)))
    print('   per refill %.0f cyc, per leaf round %.0f cyc, leaf load %.0f math %.0f per leaf round, lanes/round %.1f, leaf lanes per leaf round %.1f' % (
        tm[:, 2].sum() / max(tm[:, 3].sum(), 1), tm[:, 4].sum() / max(tm[:, 5].sum(), 1), tm[:, 6].sum() / max(tm[:, 5].sum(), 1), tm[:, 7].sum() / max(tm[:, 5].sum(), 1),
        tm[:, 8].sum() / tm[:, 1].sum(), tm[:, 9].sum() / max(tm[:, 5].sum(), 1)))
