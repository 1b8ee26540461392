import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from rust_raytracer_amd import meshes, native
md = meshes.ensure_model_dir()
host = native.HostScene.load('/root/repo/scenes/reflective_bunnies.json', md)
host.resize_screen(1920, 1080)
dev = native.DeviceScene(host.desc(), 0)
m, w, h = host.camera(); dev.set_camera(m, w, h)
side = torch.cuda.Stream(); torch.cuda.set_stream(side); stream = side.cuda_stream
n = int(sys.argv[1])
buf = torch.zeros(dev.shard_bytes(n), dtype=torch.uint8, device='cuda')
worst = 0
for r in range(n):
    for _ in range(3):
        dev.render_device(buf.data_ptr(), shard_index=r, shard_count=n, stream=stream); dev.last_ray_counts()
    t = []
    for _ in range(5):
        dev.render_device(buf.data_ptr(), shard_index=r, shard_count=n, stream=stream); t.append(dev.last_render_ms())
    worst = max(worst, min(t))
print('shards', n, 'list blocks/CU', os.environ.get('RT_LIST_BLOCKS_PER_CU', '4'), 'traverse blocks/CU', os.environ.get('RT_BLOCKS_PER_CU', 'max'), 'slowest shard %.3f ms' % worst)
