"""The BVH walk with and without a look at the ray's key every RT_WALK_POLL rounds (what other pairs of the ray have found
bounds this pair): kernel times and the walk's own work counters.  Usage: python scratch/walk_poll.py <workload> [views]"""
import sys, os, subprocess
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 3:                                        # child: one setting
    import numpy as np, torch
    import bench
    from rust_raytracer_amd import native, synthetic, meshes
    name, views = sys.argv[1], int(sys.argv[2])
    scene_file, width, height, max_depth = bench.WORKLOADS[name]
    if scene_file.startswith('synthetic:'):
        opts = dict(kv.split('=') for kv in scene_file[len('synthetic:'):].split(','))
        d = '/tmp/walk_poll_' + name
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, name + '.json')
        open(path, 'w').write(synthetic.instance_grid_scene(d, int(opts['grid']), int(opts['meshes']), int(opts['level']), width, height))
    else:
        d = meshes.ensure_model_dir()
        path = os.path.join('/root/repo/scenes', scene_file)
    host = native.HostScene.load(path, d)
    host.resize_screen(width, height)
    dev = native.DeviceScene(host.desc(), 0)
    m, w, h = host.camera()
    dev.set_views(np.tile(np.asarray(m, dtype=np.float32).reshape(1, 16), (views, 1)), w, h)
    buf = torch.zeros(views * w * h * 3, dtype=torch.uint8, device='cuda')
    dev.set_kernel_timing(True)
    for _ in range(3):
        dev.render_device(buf.data_ptr(), max_depth=max_depth); c = dev.last_ray_counts()
    ms = dev.last_kernel_ms()
    print('RT_WALK_POLL=%-4s %s x%d: node visits %d triangle tests %d | %s | checksum %d' % (
        os.environ.get('RT_WALK_POLL', '-'), name, views, c[13], c[14], ' '.join('%s %.3f ms' % (k, v[0] / views) for k, v in ms.items()),
        int(buf.to(torch.int64).sum())), flush=True)
else:
    for poll in ('0', '4', '16', '64'):
        subprocess.run([sys.executable, __file__, sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '1', 'child'], env=dict(os.environ, RT_WALK_POLL=poll), check=True)
