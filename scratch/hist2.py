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
L.rt_debug_wave_timing.argtypes = [C.c_void_p, C.c_size_t]
names = ['cycles', 'rounds', 'nodefetch', 'boxtest', 'pop', 'leafrounds', 'leafload', 'leafmath', 'lanes_sum', 'service']
for level in (0, 1, 2, 5, 9):
    os.environ['RT_DEBUG_STOP_LEVEL'] = str(level)
    dev.render(max_depth=10)
    tm = np.zeros((8192, 10), dtype=np.uint64)
    L.rt_debug_wave_timing(tm.ctypes.data, tm.size)
    tm = tm[tm[:, 1] > 0].astype(np.float64)
    tot = tm[:, 0].sum()
    print('level', level, 'waves', len(tm), 'cycles mean %.0f max %.0f rounds mean %.1f max %.0f lanes/round %.1f' % (tm[:, 0].mean(), tm[:, 0].max(), tm[:, 1].mean(), tm[:, 1].max(), tm[:, 8].sum() / tm[:, 1].sum()))
    print('   service %.1f%% of wave time; leaf rounds %.0f%% of rounds, %.1f lanes and %.0f cycles per leaf round (%.1f%% of wave time)' % (
        100 * tm[:, 9].sum() / tot, 100 * tm[:, 5].sum() / tm[:, 1].sum(), tm[:, 6].sum() / max(tm[:, 5].sum(), 1), tm[:, 7].sum() / max(tm[:, 5].sum(), 1), 100 * tm[:, 7].sum() / tot))
