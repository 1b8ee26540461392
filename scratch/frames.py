# A few frames of a bench.py workload, one at a time, for rocprofv3 passes: python3 scratch/frames.py <workload> [frames]
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
for _ in range(n):
    r = dev.render(max_depth=max_depth)
print(name, int(r['counters'][0]), 'rays', dev.last_trace_ms(), 'ms')
