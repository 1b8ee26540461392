import sys, os
sys.path.insert(0, '.')
import numpy as np
from rust_raytracer_amd import meshes, native, sharding
import torch
md = meshes.ensure_model_dir(levels={'bunny': 4, 'santa': 4, 'sphere_high': 3}) if 'levels' in meshes.ensure_model_dir.__code__.co_varnames else meshes.ensure_model_dir()
host = native.HostScene.load('scenes/reflective_bunnies.json', md)
W, H = 333, 245
host.resize_screen(W, H)
dev = native.DeviceScene(host.desc(), 0)
m, w, h = host.camera(); dev.set_camera(m, w, h)
whole = dev.render(channels=3)["rgb"]
again = dev.render(channels=3)["rgb"]
print('whole vs again differing pixels', int((whole != again).any(axis=2).sum()))
side = torch.cuda.Stream(); torch.cuda.set_stream(side); stream = side.cuda_stream
for shards in (2, 3, 8, 16):
    nbytes = dev.shard_bytes(shards)
    for rep in range(3):
        bad = 0
        for r in range(shards):
            part = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')
            dev.render_device(part.data_ptr(), shard_index=r, shard_count=shards, stream=stream)
            torch.cuda.synchronize()
            got = part.cpu().numpy(); want = sharding.extract_shard(whole, r, shards)
            d = np.nonzero(got != want)[0]
            if len(d):
                bad += 1
                print('  shards', shards, 'rep', rep, 'shard', r, 'differing bytes', len(d), 'first at', d[:6], 'got', got[d[:6]], 'want', want[d[:6]])
        print("shards", shards, "rep", rep, "bad shards", bad)
