# Every workload of bench.py at its full size: device against oracle, every pixel (ids, flags, ray counts exact; RGB8).
import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import bench
from oracle import orc
from rust_raytracer_amd import meshes, native, synthetic
import test_gpu_parity as T
names = sys.argv[1:] or sorted(bench.WORKLOADS)
for name in names:
    scene_file, width, height, max_depth = bench.WORKLOADS[name]
    t0 = time.time()
    if scene_file.startswith('synthetic:'):
        opts = dict(kv.split('=') for kv in scene_file[len('synthetic:'):].split(','))
        d = '/tmp/full_size_grid'
        os.makedirs(d, exist_ok=True)
        s = orc.OracleScene.from_text(synthetic.instance_grid_scene(d, int(opts['grid']), int(opts['meshes']), int(opts['level']), width, height), d)
    else:
        d = meshes.ensure_model_dir('/tmp/full_size_models')
        s = orc.OracleScene.from_file('/root/repo/scenes/' + scene_file, d)
        s.resize(width, height)
    print(name, 'loaded in %.0f s' % (time.time() - t0), flush=True)
    got, want = T.render_both(s, max_depth=max_depth)
    exact, within = T.assert_parity(got, want)
    print('%s %dx%d depth %d: %d rays, ids/flags/counters exact, RGB8 identical on %.5f of the pixels, within 1 LSB on %.5f (%.0f s)'
          % (name, width, height, max_depth, int(got['counters'][0]), exact, within, time.time() - t0), flush=True)
