import sys, os, time
sys.path.insert(0, '.')
import numpy as np, torch
from rust_raytracer_amd import meshes, native
md = meshes.ensure_model_dir()
host = native.HostScene.load('scenes/reflective_bunnies.json', md)
host.resize_screen(1920, 1080)
dev = native.DeviceScene(host.desc(), 0)
m, w, h = host.camera(); dev.set_camera(m, w, h)
side = torch.cuda.Stream(); torch.cuda.set_stream(side); stream = side.cuda_stream
base = None
for n in (1, 2, 4, 8):
    nbytes = dev.shard_bytes(n)
    buf = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')
    worst = 0.0; rays = 0
    for r in range(n):
        for _ in range(3):
            dev.render_device(buf.data_ptr(), shard_index=r, shard_count=n, stream=stream)
            dev.last_ray_counts()
        t = []
        for _ in range(5):
            dev.render_device(buf.data_ptr(), shard_index=r, shard_count=n, stream=stream)
            t.append(dev.last_render_ms())
        worst = max(worst, min(t))
        if r == 0: rays0 = int(dev.last_ray_counts()[0])
    base = base or worst
    print('shards', n, 'slowest shard %.3f ms' % worst, 'speedup %.2f' % (base / worst), 'efficiency %.0f%%' % (100 * base / worst / n))
