"""A lone 1080p frame (and a lone 1/8 shard) enqueued launch by launch against the same frame replayed as a HIP graph
(torch.cuda.CUDAGraph capturing rt_render_device's launches on the capture stream)."""
import sys, os, time
sys.path.insert(0, '.')
import numpy as np, torch
from rust_raytracer_amd import meshes, native
md = meshes.ensure_model_dir()
host = native.HostScene.load('scenes/reflective_bunnies.json', md)
host.resize_screen(1920, 1080)
m, w, h = host.camera()
for shards in (1, 8):
    dev = native.DeviceScene(host.desc(), 0)
    dev.set_camera(m, w, h)
    s = torch.cuda.Stream()
    buf = torch.zeros(dev.shard_bytes(shards), dtype=torch.uint8, device='cuda')
    def frame():
        dev.render_device(buf.data_ptr(), shard_index=0, shard_count=shards, stream=s.cuda_stream)
    for _ in range(3):
        frame(); torch.cuda.synchronize(); dev.last_ray_counts()
    want = buf.clone()
    plain = []
    for _ in range(30):
        t0 = time.perf_counter(); frame(); torch.cuda.synchronize(); plain.append(time.perf_counter() - t0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        frame()
    torch.cuda.synchronize()
    buf.zero_()
    g.replay(); torch.cuda.synchronize()
    same = bool(torch.equal(buf, want))
    graph = []
    for _ in range(30):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); graph.append(time.perf_counter() - t0)
    print('1/%d of the frame: launch by launch %.3f ms (min %.3f), as a graph %.3f ms (min %.3f), same bytes %s'
          % (shards, np.median(plain) * 1e3, min(plain) * 1e3, np.median(graph) * 1e3, min(graph) * 1e3, same), flush=True)
    del g
    dev.close()
