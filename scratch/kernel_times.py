# Per-kernel milliseconds per frame (HIP events, one frame at a time) for bench.py workloads: python scratch/kernel_times.py [workload ...]
import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from rust_raytracer_amd import meshes, native, synthetic
views = int(os.environ.get('VIEWS', '1'))
names = sys.argv[1:] or ['reflective_bunnies_1080p']
for name in names:
    scene_file, width, height, max_depth = bench.WORKLOADS[name]
    if scene_file.startswith('synthetic:'):
        opts = dict(kv.split('=') for kv in scene_file[len('synthetic:'):].split(','))
        d = '/tmp/kernel_times_grid'
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, name + '.json')
        open(path, 'w').write(synthetic.instance_grid_scene(d, int(opts['grid']), int(opts['meshes']), int(opts['level']), width, height))
        host = native.HostScene.load(path, d)
    else:
        host = native.HostScene.load('/root/repo/scenes/' + scene_file, meshes.ensure_model_dir())
    host.resize_screen(width, height)
    dev = native.DeviceScene(host.desc(), 0)
    m, w, h = host.camera()
    # the views of a pass: the scene's own camera and drags of it, 12 pixels apart, as in bench.py (SAME_CAMERA=1: copies of one camera)
    cams = [np.asarray(m, dtype=np.float32).reshape(16)]
    position0, orientation0 = host.camera_pose()
    for k in range(1, views):
        if os.environ.get('SAME_CAMERA') == '1': cams.append(cams[0]); continue
        host.set_camera_pose(position0, orientation0)
        step = 12.0 * ((k + 1) // 2)
        host.turn_camera(0.0, 0.0, step if k % 2 else -step, 0.0)
        cams.append(np.asarray(host.camera()[0], dtype=np.float32).reshape(16))
    host.set_camera_pose(position0, orientation0)
    dev.set_views(np.stack(cams), w, h)
    buf = torch.zeros(views * w * h * 3, dtype=torch.uint8, device='cuda')
    for shards in (1, 8):
        for _ in range(3):
            dev.render_device(buf.data_ptr(), max_depth=max_depth, shard_count=shards); dev.last_ray_counts()
        dev.set_kernel_timing(True)
        tot = {}
        frame = []
        n = 5
        for _ in range(n):
            dev.render_device(buf.data_ptr(), max_depth=max_depth, shard_count=shards)
            for k, (ms, c) in dev.last_kernel_ms().items():
                tot[k] = tot.get(k, 0.0) + ms / n / views
            frame.append(dev.last_render_ms())
        dev.set_kernel_timing(False)
        plain = []
        for _ in range(n):
            dev.render_device(buf.data_ptr(), max_depth=max_depth, shard_count=shards); dev.last_ray_counts()
            plain.append(dev.last_render_ms())
        print('%s 1/%d: frame %.3f ms (timed mode %.3f) rays %d | %s' % (name, shards, float(np.median(plain)) / views, float(np.median(frame)) / views, int(dev.last_ray_counts()[0]) // views,
              ' '.join('%s %.3f' % (k, v) for k, v in tot.items())), flush=True)
    dev.close(); host.close()
