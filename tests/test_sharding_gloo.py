"""The N > 1 path of bench.py on CPU: two gloo ranks shard a frame by tiles, all-gather, assemble.
The per-rank "renderer" is the numpy statement of the shard layout (rust_raytracer_amd/sharding.py);
the GPU tests check the HIP kernels against that same statement."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rust_raytracer_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, width, height, channels, out_dir, views=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(1234)                      # every rank knows the whole reference frame
    frame = rng.integers(0, 256, size=(views, height, width, channels), dtype=np.uint8)

    def render_shard(out):
        out.copy_(torch.from_numpy(np.concatenate([sharding.extract_shard(frame[v], rank, world) for v in range(views)])))

    def assemble(gathered, out):
        out.copy_(torch.from_numpy(sharding.assemble(gathered.numpy(), width, height, world, channels, views).reshape(-1)))

    sf = sharding.ShardedFrame(width, height, channels, rank, world, "cpu", render_shard, assemble, views=views)
    got = sf.step().numpy().reshape(views, height, width, channels)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), got)
    ok = np.array_equal(got, frame)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    assert flag.item() == 1


@pytest.mark.parametrize("width,height,channels,views", [(1920, 1080, 3, 1), (333, 117, 4, 1), (333, 117, 3, 3)])
def test_two_ranks_assemble_the_frame(tmp_path, width, height, channels, views):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), width, height, channels, str(tmp_path), views), nprocs=world, join=True)
    a = np.load(tmp_path / "rank0.npy")
    b = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b)


@pytest.mark.parametrize("shards", [1, 2, 3, 4, 8])
def test_extract_and_assemble_roundtrip(shards):
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, size=(117, 333, 3), dtype=np.uint8)
    parts = [sharding.extract_shard(frame, r, shards) for r in range(shards)]
    assert all(len(p) == sharding.shard_bytes(333, 117, shards) for p in parts) or shards == 1
    if shards > 1:
        assert np.array_equal(sharding.assemble(np.concatenate(parts), 333, 117, shards), frame)
