# Random scenes of the fuzz generator: tile shards and several views per pass against whole single frames (device only).
import sys, os, tempfile
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from rust_raytracer_amd import native, synthetic
a, b = int(sys.argv[1]), int(sys.argv[2])
d = tempfile.mkdtemp()
side = torch.cuda.Stream(); torch.cuda.set_stream(side)
bad = []
for seed in range(a, b):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(20, 140)), int(rng.integers(20, 100))
    path = os.path.join(d, 's%d.json' % seed)
    open(path, 'w').write(synthetic.random_scene(d, seed, w, h))
    host = native.HostScene.load(path, d)
    dev = native.DeviceScene(host.desc(), 0)
    try:
        views = int(rng.integers(1, 4)); shards = int(rng.choice([2, 3, 5, 8, 16]))
        mats, singles = [], []
        for v in range(views):
            host.turn_camera(0.0, 0.0, float(rng.uniform(-30, 30)), float(rng.uniform(-10, 10)))
            m, ww, hh = host.camera()
            mats.append(np.asarray(m, dtype=np.float32).reshape(16))
            dev.set_camera(m, ww, hh)
            singles.append(dev.render(max_depth=5)['rgb'])
        dev.set_views(np.stack(mats), ww, hh)
        nbytes = dev.shard_bytes(shards)
        gathered = torch.zeros(shards * views * nbytes, dtype=torch.uint8, device='cuda')
        for r in range(shards):
            dev.render_device(gathered[r * views * nbytes:].data_ptr(), max_depth=5, shard_index=r, shard_count=shards, stream=side.cuda_stream)
            dev.last_ray_counts()
        frames = torch.zeros(views * hh * ww * 3, dtype=torch.uint8, device='cuda')
        dev.assemble_shards(gathered.data_ptr(), frames.data_ptr(), shards, stream=side.cuda_stream)
        torch.cuda.synchronize()
        got = frames.cpu().numpy().reshape(views, hh, ww, 3)
        if not all(np.array_equal(got[v], singles[v]) for v in range(views)):
            bad.append(seed); print('seed', seed, 'FAILED', ww, hh, views, shards, flush=True)
    finally:
        dev.close(); host.close(); os.remove(path)
    if seed % 200 == 0: print('at seed', seed, flush=True)
print('seeds', a, b, 'failures', bad)
