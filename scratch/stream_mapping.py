# Does the time of three passes in flight depend on which streams of torch's pool carry them?
import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from rust_raytracer_amd import meshes, native
md = meshes.ensure_model_dir()
host = native.HostScene.load('/root/repo/scenes/reflective_bunnies.json', md)
host.resize_screen(1920, 1080)
m, w, h = host.camera()
shards, depth = 8, 3
devs = [native.DeviceScene(host.desc(), 0) for _ in range(depth)]
for d in devs: d.set_camera(m, w, h)
nbytes = devs[0].shard_bytes(shards)
bufs = [torch.zeros(nbytes, dtype=torch.uint8, device='cuda') for _ in range(depth)]
pool = [torch.cuda.Stream() for _ in range(32)]
print('pool handles', [hex(s.cuda_stream)[-6:] for s in pool[:8]], flush=True)
for pick in ((0, 1, 2), (1, 2, 3), (2, 3, 4), (3, 4, 5), (0, 4, 8), (0, 1, 4), (0, 2, 5), (5, 6, 7)):
    streams = [pool[i] for i in pick]
    def frame(k):
        s = k % depth
        devs[s].render_device(bufs[s].data_ptr(), shard_index=0, shard_count=shards, stream=streams[s].cuda_stream)
    for k in range(6):
        frame(k); torch.cuda.synchronize()
    res = []
    for rep in range(3):
        K = 30
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K): frame(k)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / K * 1e3)
    print('streams', pick, 'ms per frame', ' '.join('%.3f' % x for x in res), flush=True)
