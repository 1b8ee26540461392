# Time per ray as the frame grows (a stand-in for rendering several frames in one chain of launches).
import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from rust_raytracer_amd import meshes, native
md = meshes.ensure_model_dir()
host = native.HostScene.load('/root/repo/scenes/reflective_bunnies.json', md)
for (W, H) in ((1920, 1080), (2716, 1528), (3840, 2160), (5432, 3056)):
    host.resize_screen(W, H)
    m, w, h = host.camera()
    for depth in (1, 3):
        devs = [native.DeviceScene(host.desc(), 0) for _ in range(depth)]
        streams = [torch.cuda.Stream() for _ in range(depth)]
        for d in devs: d.set_camera(m, w, h)
        bufs = [torch.zeros(w * h * 3, dtype=torch.uint8, device='cuda') for _ in range(depth)]
        def frame(k):
            s = k % depth
            devs[s].render_device(bufs[s].data_ptr(), stream=streams[s].cuda_stream)
        for k in range(3 * depth):
            frame(k); torch.cuda.synchronize()
        rays = int(devs[0].last_ray_counts()[0])
        K = 12
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K): frame(k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print('%dx%d in flight %d: %.3f ms per frame, %d rays, %.0f Mrays/s' % (W, H, depth, dt * 1e3, rays, rays / dt / 1e6), flush=True)
        for d in devs: d.close()
