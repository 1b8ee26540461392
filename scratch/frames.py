# A few frames of a bench.py workload, one at a time, for rocprofv3 passes: python3 scratch/frames.py <workload> [frames]
# VIEWS=n: passes of n frames (n cameras 12 pixels of drag apart, as in bench.py) instead of single frames
import sys, os
sys.path.insert(0, '/root/repo')
import bench
from rust_raytracer_amd import meshes, native, synthetic
name = sys.argv[1] if len(sys.argv) > 1 else 'reflective_bunnies_1080p'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
scene_file, width, height, max_depth = bench.WORKLOADS[name]
if scene_file.startswith('synthetic:'):
    opts = dict(kv.split('=') for kv in scene_file[len('synthetic:'):].split(','))
    d = '/tmp/frames_grid'
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, name + '.json')
    open(path, 'w').write(synthetic.instance_grid_scene(d, int(opts['grid']), int(opts['meshes']), int(opts['level']), width, height))
    host = native.HostScene.load(path, d)
elif name.endswith('_sphere_standin'):
    d = '/tmp/frames_sphere'
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, 'bunny.obj'), 'w').write(meshes.obj_text(*meshes.uv_sphere(5)))
    host = native.HostScene.load('/root/repo/scenes/' + scene_file, d)
else:
    host = native.HostScene.load('/root/repo/scenes/' + scene_file, meshes.ensure_model_dir())
host.resize_screen(width, height)
dev = native.DeviceScene(host.desc(), 0)
m, w, h = host.camera(); dev.set_camera(m, w, h)
views = int(os.environ.get('VIEWS', '1'))
if views > 1:
    import numpy as np, torch
    position0, orientation0 = host.camera_pose()
    cams = [np.asarray(m, dtype=np.float32).reshape(16)]
    for k in range(1, views):
        host.set_camera_pose(position0, orientation0)
        step = 12.0 * ((k + 1) // 2)
        host.turn_camera(0.0, 0.0, step if k % 2 else -step, 0.0)
        cams.append(np.asarray(host.camera()[0], dtype=np.float32).reshape(16))
    dev.set_views(np.stack(cams), w, h)
    buf = torch.zeros(views * w * h * 3, dtype=torch.uint8, device='cuda')
    for _ in range(n):
        dev.render_device(buf.data_ptr(), max_depth=max_depth)
        c = dev.last_ray_counts()
    print(name, int(c[0]), 'rays in a pass of', views, 'views', dev.last_trace_ms(), 'ms')
else:
    for _ in range(n):
        r = dev.render(max_depth=max_depth)
    print(name, int(r['counters'][0]), 'rays', dev.last_trace_ms(), 'ms')
