import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from rust_raytracer_amd import meshes, native
md = meshes.ensure_model_dir()
host = native.HostScene.load('/root/repo/scenes/reflective_bunnies.json', md)
host.resize_screen(1920, 1080)
m, w, h = host.camera()
for shards in (1, 2, 4, 8):
    for depth in (1, 2, 3):
        devs = [native.DeviceScene(host.desc(), 0) for _ in range(depth)]
        streams = [torch.cuda.Stream() for _ in range(depth)]
        for d in devs: d.set_camera(m, w, h)
        nbytes = devs[0].shard_bytes(shards)
        bufs = [torch.zeros(nbytes, dtype=torch.uint8, device='cuda') for _ in range(depth)]
        def frame(k):
            s = k % depth
            devs[s].render_device(bufs[s].data_ptr(), shard_index=0, shard_count=shards, stream=streams[s].cuda_stream)
        for k in range(3 * depth):
            frame(k); torch.cuda.synchronize()
            for d in devs[:1]: d.last_ray_counts()
        K = 30
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K): frame(k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print('shards', shards, 'frames in flight', depth, 'ms per frame %.3f' % (dt * 1e3))
        for d in devs: d.close()
