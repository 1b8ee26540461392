# RCCL with one rank: the gather of bench.py's N > 1 path (all_gather_into_tensor issued under three pass streams, between
# the library's launches on those streams) on the real backend, as far as one GPU allows.
import sys, os, time
sys.path.insert(0, '/root/repo')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29611')
import numpy as np, torch, torch.distributed as dist
from rust_raytracer_amd import meshes, native, sharding
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
md = meshes.ensure_model_dir()
host = native.HostScene.load('/root/repo/scenes/reflective_bunnies.json', md)
host.resize_screen(1920, 1080)
m, w, h = host.camera()
views, depth = 4, 3
streams = [torch.cuda.Stream() for _ in range(depth)]
devs = [native.DeviceScene(host.desc(), 0) for _ in range(depth)]
for dv in devs: dv.set_views(np.tile(np.asarray(m, dtype=np.float32).reshape(1, 16), (views, 1)), w, h)
n = devs[0].shard_bytes(2) * views                       # rendered as shard 0 of 2, so that the tiled path and the gather are used
shard = [torch.zeros(n, dtype=torch.uint8, device='cuda') for _ in range(depth)]
gathered = [torch.zeros(n, dtype=torch.uint8, device='cuda') for _ in range(depth)]
def step(k):
    s = k % depth
    with torch.cuda.stream(streams[s]):
        devs[s].render_device(shard[s].data_ptr(), shard_index=0, shard_count=2, stream=streams[s].cuda_stream)
        dist.all_gather_into_tensor(gathered[s], shard[s])
for k in range(6): step(k); torch.cuda.synchronize()
ref = gathered[0].clone()
for g in gathered: g.zero_()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 12
for k in range(K): step(k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (K * views)
ok = all(torch.equal(g, ref) for g in gathered) and bool(ref.any())
print('gathered buffers equal the shard and each other:', ok, '- %.3f ms per half frame' % (dt * 1e3))
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
