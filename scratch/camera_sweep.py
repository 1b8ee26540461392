# rt_traverse time of single frames at the drag cameras bench.py uses, and of passes made of chosen subsets: python scratch/camera_sweep.py
import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from rust_raytracer_amd import meshes, native
host = native.HostScene.load('/root/repo/scenes/reflective_bunnies.json', meshes.ensure_model_dir())
host.resize_screen(1920, 1080)
m, w, h = host.camera()
cams = [np.asarray(m, dtype=np.float32).reshape(16)]
position0, orientation0 = host.camera_pose()
for k in range(1, 16):
    host.set_camera_pose(position0, orientation0)
    step = 12.0 * ((k + 1) // 2)
    host.turn_camera(0.0, 0.0, step if k % 2 else -step, 0.0)
    cams.append(np.asarray(host.camera()[0], dtype=np.float32).reshape(16))
dev = native.DeviceScene(host.desc(), 0)
def run(sel):
    dev.set_views(np.stack([cams[i] for i in sel]), w, h)
    buf = torch.zeros(len(sel) * w * h * 3, dtype=torch.uint8, device='cuda')
    for _ in range(3):
        dev.render_device(buf.data_ptr(), max_depth=10)
        try: dev.last_ray_counts()
        except native.RtError: pass
    dev.set_kernel_timing(True)
    tot = {}
    for _ in range(3):
        dev.render_device(buf.data_ptr(), max_depth=10)
        for k, (ms, c) in dev.last_kernel_ms().items(): tot[k] = tot.get(k, 0.0) + ms / 3
    dev.set_kernel_timing(False)
    c = dev.last_ray_counts()
    print(sel, 'rays', int(c[0]), ' '.join('%s %.3f' % kv for kv in tot.items()), 'node visits', int(c[5]), 'tri tests', int(c[6]), flush=True)
for k in range(8): run([k])
run([0, 1]); run([0, 1, 2]); run([0, 1, 2, 3]); run([2, 3]); run([4, 5, 6, 7]); run([0, 0, 0, 0])
