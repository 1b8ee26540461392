# One rank's time per frame for its 1/n shard, with `views` frames per pass and three passes in flight
# (no gather: what tile sharding can reach on n GPUs).
import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from rust_raytracer_amd import meshes, native
md = meshes.ensure_model_dir()
host = native.HostScene.load('/root/repo/scenes/reflective_bunnies.json', md)
host.resize_screen(1920, 1080)
m, w, h = host.camera()
base = None
for shards in (1, 2, 4, 8):
    for views in sorted({1, shards, 2 * shards, min(32, 4 * shards)}):
        depth = 3
        devs = [native.DeviceScene(host.desc(), 0) for _ in range(depth)]
        streams = [torch.cuda.Stream() for _ in range(depth)]
        for d in devs: d.set_views(np.tile(np.asarray(m, dtype=np.float32).reshape(1, 16), (views, 1)), w, h)
        nbytes = devs[0].shard_bytes(shards) * views
        bufs = [torch.zeros(nbytes, dtype=torch.uint8, device='cuda') for _ in range(depth)]
        def frame(k):
            s = k % depth
            devs[s].render_device(bufs[s].data_ptr(), shard_index=0, shard_count=shards, stream=streams[s].cuda_stream)
        for k in range(2 * depth):
            frame(k); torch.cuda.synchronize()
        devs[0].last_ray_counts()
        K = max(6, 48 // views)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K): frame(k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (K * views)
        if base is None or (shards == 1 and dt < base): base = dt
        print('shards %d views per pass %2d: %.3f ms per frame, efficiency against the best one-GPU rate %.0f %%' % (shards, views, dt * 1e3, 100 * base / (dt * shards)), flush=True)
        for d in devs: d.close()
