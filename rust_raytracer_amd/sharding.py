"""Tile sharding of a frame over ranks: the layout contract of rt_render_device / rt_assemble_shards
(include/rt_hip.h) restated in numpy, plus the per-rank driver used by bench.py.

A frame is cut into 32x32-pixel tiles numbered row-major; rank r of n owns the tiles t with
t % n == r (the two-dimensional version of the row interlacing in
web_app/wasm/src/renderer.rs:68-76 and web_app/src/main/model.ts:115-129).  A rank's buffer holds
its tiles back to back, pixels row-major inside a tile, zero outside the frame; every rank's buffer
has the same size so that one all-gather moves them.
"""
from __future__ import annotations

import numpy as np

from .abi import RT_TILE_DIM as T


def tile_grid(width: int, height: int):
    return (width + T - 1) // T, (height + T - 1) // T


def tiles_per_shard(width: int, height: int, shard_count: int) -> int:
    tx, ty = tile_grid(width, height)
    return (tx * ty + shard_count - 1) // shard_count


def shard_bytes(width: int, height: int, shard_count: int, channels: int = 3) -> int:
    if shard_count <= 1:
        return width * height * channels
    return tiles_per_shard(width, height, shard_count) * T * T * channels


def extract_shard(frame: np.ndarray, shard_index: int, shard_count: int) -> np.ndarray:
    """The buffer rt_render_device(shard_index, shard_count) produces, cut out of a full frame [H, W, C]."""
    h, w, c = frame.shape
    tx, ty = tile_grid(w, h)
    n = tiles_per_shard(w, h, shard_count)
    out = np.zeros((n, T, T, c), dtype=frame.dtype)
    for k in range(n):
        t = shard_index + k * shard_count
        if t >= tx * ty:
            break
        y0, x0 = (t // tx) * T, (t % tx) * T
        block = frame[y0:y0 + T, x0:x0 + T]
        out[k, :block.shape[0], :block.shape[1]] = block
    return out.reshape(-1)


def assemble(gathered: np.ndarray, width: int, height: int, shard_count: int, channels: int = 3, views: int = 1) -> np.ndarray:
    """What rt_assemble_shards computes: rank-major gathered shard buffers -> frame [H, W, C].  With several views
    (rt_scene_set_views) a rank's buffer holds its shard of every view, view after view, and the result is
    [views, H, W, C]."""
    tx, ty = tile_grid(width, height)
    n = tiles_per_shard(width, height, shard_count)
    g = np.asarray(gathered).reshape(shard_count, views, n, T, T, channels)
    frames = np.zeros((views, height, width, channels), dtype=g.dtype)
    for t in range(tx * ty):
        y0, x0 = (t // tx) * T, (t % tx) * T
        hh, ww = min(T, height - y0), min(T, width - x0)
        frames[:, y0:y0 + hh, x0:x0 + ww] = g[t % shard_count, :, t // shard_count, :hh, :ww]
    return frames if views > 1 else frames[0]


class ShardedFrame:
    """One rank's part of a tile-sharded frame: render own tiles, all-gather, assemble.

    `render_shard(out_tensor)` fills this rank's shard buffer, `assemble_fn(gathered, frame)` turns
    the gathered buffers into the frame; both are injected so that the same driver runs over the
    HIP library on GPUs (bench.py) and over numpy on CPU ranks (tests, gloo).  `gather_fn(gathered, shard)`
    defaults to torch.distributed.all_gather_into_tensor.
    """

    def __init__(self, width, height, channels, rank, world_size, device, render_shard, assemble_fn, gather_fn=None,
                 views=1, always_gather=False):
        import torch
        self.torch = torch
        self.rank, self.world = rank, world_size
        self.width, self.height, self.channels = width, height, channels
        nbytes = shard_bytes(width, height, world_size, channels) * views      # `views` frames travel together
        self.shard = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        self.frame = torch.zeros(views * height * width * channels, dtype=torch.uint8, device=device)
        # always_gather: a single rank goes through the exchange step as well (its "shard" is the whole frame, the gather
        # of one buffer a copy made by the collective library) - the way to run the collective branch on a one-GPU box
        self.always_gather = always_gather
        self.gathered = torch.zeros(nbytes * world_size, dtype=torch.uint8, device=device) if (world_size > 1 or always_gather) else None
        self.render_shard, self.assemble_fn = render_shard, assemble_fn
        if gather_fn is None:
            import torch.distributed as dist
            gather_fn = dist.all_gather_into_tensor
        self.gather_fn = gather_fn

    def step(self):
        if self.world == 1 and not self.always_gather:
            self.render_shard(self.frame)
            return self.frame
        self.render_shard(self.shard)
        self.gather_fn(self.gathered, self.shard)                   # the frame's one exchange step
        self.assemble_fn(self.gathered, self.frame)
        return self.frame
