import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from rust_raytracer_amd import meshes, native
md = meshes.ensure_model_dir()
host = native.HostScene.load('/root/repo/scenes/%s.json' % (sys.argv[2] if len(sys.argv) > 2 else 'reflective_bunnies'), md)
host.resize_screen(1920, 1080)
dev = native.DeviceScene(host.desc(), 0)
m, w, h = host.camera(); dev.set_camera(m, w, h)
side = torch.cuda.Stream(); torch.cuda.set_stream(side); stream = side.cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
buf = torch.zeros(dev.shard_bytes(n), dtype=torch.uint8, device='cuda')
for _ in range(5):
    dev.render_device(buf.data_ptr(), shard_index=0, shard_count=n, stream=stream); dev.last_ray_counts()
