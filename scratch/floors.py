"""Where the BVH walk's launch time goes, depth by depth (RT_STATS build: `make -C rust_raytracer_amd/csrc stats`).
Usage: python scratch/floors.py [shard_count]   (1 = whole frame)"""
import sys, os, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from rust_raytracer_amd import meshes, native
native.LIB_PATH = os.path.abspath('scratch/librt_hip_stats.so')
md = meshes.ensure_model_dir()
host = native.HostScene.load('scenes/reflective_bunnies.json', md)
host.resize_screen(1920, 1080)
dev = native.DeviceScene(host.desc(), 0)
k = int(os.environ.get('CAMERA_K', '0'))                  # drag camera k of bench.py
if k:
    step = 12.0 * ((k + 1) // 2)
    host.turn_camera(0.0, 0.0, step if k % 2 else -step, 0.0)
m, w, h = host.camera(); dev.set_camera(m, w, h)
side = torch.cuda.Stream(); torch.cuda.set_stream(side); stream = side.cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
buf = torch.zeros(dev.shard_bytes(n), dtype=torch.uint8, device='cuda')
L = native.lib()
L.rt_debug_wave_timing.argtypes = [C.c_void_p, C.c_size_t]
for level in [int(x) for x in os.environ.get('LEVELS', '0,1,2,3,4,5,6,7,8,9,10').split(',')]:
    os.environ['RT_DEBUG_STOP_LEVEL'] = str(level)
    for _ in range(2):
        dev.render_device(buf.data_ptr(), shard_index=0, shard_count=n, stream=stream); torch.cuda.synchronize()
    tm = np.zeros((8192, 10), dtype=np.uint64)
    L.rt_debug_wave_timing(tm.ctypes.data, tm.size)
    allw = tm[tm[:, 0] > 0].astype(np.float64)
    act = allw[allw[:, 8] > 0]
    if len(act) == 0:
        print('level', level, 'no wave had work'); continue
    t0 = allw[:, 3].min()
    span = (allw[:, 4].max() - t0) / 100.0            # wall clock ticks of 10 ns
    start_last = (allw[:, 3].max() - t0) / 100.0
    longest = act[np.argmax(act[:, 0])]
    work = act[:, 0] - act[:, 2]
    print('   per round: service %.0f cyc (in %.0f%% of cycles), leaf phase %.0f cyc in %.0f%% of the rounds, the rest (node step, ballots) %.0f cyc'
          % (act[:, 9].sum() / act[:, 1].sum(), 100 * act[:, 9].sum() / work.sum(), act[:, 7].sum() / max(act[:, 5].sum(), 1), 100 * act[:, 5].sum() / act[:, 1].sum(),
             (work.sum() - act[:, 9].sum() - act[:, 7].sum()) / act[:, 1].sum()))
    print('level %2d | waves %4d of %4d with work | span %.1f us, last wave starts at %.1f us | prefix %.0f cyc mean | cycles mean %.0f max %.0f | '
          'rounds mean %.1f max %.0f | longest wave: %.0f rounds, %.0f lane-steps, service %.0f%%, %.0f cyc/round | lane-steps total %.0f'
          % (level, len(act), len(allw), span, start_last, allw[:, 2].mean(), act[:, 0].mean(), act[:, 0].max(), act[:, 1].mean(), act[:, 1].max(),
             longest[1], longest[8], 100 * longest[9] / longest[0], longest[0] / max(longest[1], 1), act[:, 8].sum()))
