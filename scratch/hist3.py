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
side = torch.cuda.Stream(); torch.cuda.set_stream(side); stream = side.cuda_stream
n = 8
buf = torch.zeros(dev.shard_bytes(n), dtype=torch.uint8, device='cuda')
L = native.lib()
L.rt_debug_wave_timing.argtypes = [C.c_void_p, C.c_size_t]
for level in (0, 1, 5, 9):
    os.environ['RT_DEBUG_STOP_LEVEL'] = str(level)
    dev.render_device(buf.data_ptr(), shard_index=0, shard_count=n, stream=stream); torch.cuda.synchronize()
    tm = np.zeros((8192, 10), dtype=np.uint64)
    L.rt_debug_wave_timing(tm.ctypes.data, tm.size)
    act = tm[tm[:, 8] > 0].astype(np.float64)
    allw = tm[tm[:, 1] > 0].astype(np.float64)
    print('level', level, 'waves with work', len(act), 'of', len(allw), 'cycles mean %.0f max %.0f | rounds mean %.1f max %.0f | lane-steps total %.0f lanes/round %.1f | service %.0f%% | leaf rounds %.0f, %.0f cyc each' % (
        act[:, 0].mean(), act[:, 0].max(), act[:, 1].mean(), act[:, 1].max(), act[:, 8].sum(), act[:, 8].sum() / act[:, 1].sum(), 100 * act[:, 9].sum() / act[:, 0].sum(), act[:, 5].sum(), act[:, 7].sum() / max(act[:, 5].sum(), 1)))
