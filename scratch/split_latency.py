"""Latency of ONE 1080p frame rendered as P tile shards on P streams at once (host waits for every frame)."""
import sys, os, time
sys.path.insert(0, '.')
import numpy as np, torch
from rust_raytracer_amd import meshes, native
md = meshes.ensure_model_dir()
host = native.HostScene.load('scenes/reflective_bunnies.json', md)
host.resize_screen(1920, 1080)
m, w, h = host.camera()
for parts in (1, 2, 3, 4, 6, 8):
    devs = [native.DeviceScene(host.desc(), 0) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for d in devs: d.set_camera(m, w, h)
    nbytes = devs[0].shard_bytes(parts)
    buf = torch.zeros(parts * nbytes, dtype=torch.uint8, device='cuda')
    def frame():
        for r, (d, s) in enumerate(zip(devs, streams)):
            d.render_device(buf[r * nbytes:(r + 1) * nbytes].data_ptr(), shard_index=r, shard_count=parts, stream=s.cuda_stream)
    for _ in range(3):
        frame(); torch.cuda.synchronize()
        for d in devs: d.last_ray_counts()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        frame()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    rays = sum(int(d.last_ray_counts()[0]) for d in devs)
    ts = np.array(ts) * 1e3
    print('parts', parts, 'latency ms median %.3f min %.3f' % (np.median(ts), ts.min()), 'rays', rays, flush=True)
    for d in devs: d.close()
