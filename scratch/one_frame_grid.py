# Three frames of a synthetic instance-grid workload of bench.py (argv[1], default instance_grid_2p6m_4k), for PMC passes.
import sys, os
sys.path.insert(0, '/root/repo')
import bench
from rust_raytracer_amd import native, synthetic
name = sys.argv[1] if len(sys.argv) > 1 else 'instance_grid_2p6m_4k'
scene_file, width, height, max_depth = bench.WORKLOADS[name]
opts = dict(kv.split('=') for kv in scene_file[len('synthetic:'):].split(','))
d = '/tmp/one_frame_grid'
os.makedirs(d, exist_ok=True)
path = os.path.join(d, name + '.json')
open(path, 'w').write(synthetic.instance_grid_scene(d, int(opts['grid']), int(opts['meshes']), int(opts['level']), width, height))
host = native.HostScene.load(path, d)
host.resize_screen(width, height)
dev = native.DeviceScene(host.desc(), 0)
m, w, h = host.camera(); dev.set_camera(m, w, h)
import torch
buf = torch.zeros(w * h * 3, dtype=torch.uint8, device='cuda')
for _ in range(3):
    dev.render_device(buf.data_ptr(), max_depth=max_depth); dev.last_ray_counts()
print(dev.last_trace_ms())
