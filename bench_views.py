#!/usr/bin/env python3
"""Throughput mode: a batch of camera views of one uploaded scene (256 poses on a circle around
reflective_bunnies, 1920x1080 each).  View v is rendered by rank v % N; there is no collective on the data
path - every rank keeps its frames in its own HBM.  One step = the whole batch.  A rank renders its views
--views-per-pass at a time (rt_scene_set_views) and keeps --passes-in-flight passes queued, each on its own scene
handle and stream.

    python bench_views.py --gpus 1 --views 256 --steps 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench_views.py --gpus N

Rank 0 prints one JSON line (same fields as bench.py where they apply).  bench.py stays the headline
benchmark; this one measures the configuration "batch of 256 camera views, view-sharded".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--views", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cpu-sample-views", type=int, default=4)
    ap.add_argument("--views-per-pass", type=int, default=16)
    ap.add_argument("--passes-in-flight", type=int, default=3)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from rust_raytracer_amd import abi, meshes, native, synthetic

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench_views.py needs an MI355X (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    scene_path = os.path.join(ROOT, "scenes", "reflective_bunnies.json")
    model_dir = meshes.ensure_model_dir(os.path.join("/tmp", "rust_raytracer_amd_models_rank%d_%d" % (rank, os.getuid())))
    host = native.HostScene.load(scene_path, model_dir)
    host.resize_screen(args.width, args.height)
    poses = synthetic.orbit_poses(args.views, centre=(0.0, 0.0, -6.0))
    mine = list(range(rank, args.views, world))
    cameras = []
    for v in mine:
        host.set_camera_pose(*poses[v])
        cameras.append(host.camera())
    per_pass = max(1, min(args.views_per_pass, len(mine) or 1))
    depth = max(1, args.passes_in_flight)
    devs = [native.DeviceScene(host.desc(), local_rank) for _ in range(depth)]
    streams = [torch.cuda.Stream() for _ in range(depth)]
    frames = torch.zeros((depth, per_pass * args.height * args.width * 3), dtype=torch.uint8, device="cuda")   # a pass's frames
    dev, handle = devs[0], streams[0].cuda_stream
    # the views of a pass (the last pass may hold fewer: a handle's buffers serve any smaller number of views)
    passes = [np.stack([np.asarray(m, dtype=np.float32).reshape(16) for m, _, _ in cameras[k:k + per_pass]])
              for k in range(0, len(cameras), per_pass)]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def batch(count_rays=False):
        total = np.zeros(5, dtype=np.int64)
        for k, mats in enumerate(passes):
            s = k % depth
            devs[s].set_views(mats, args.width, args.height)            # the matrices travel with the launch
            devs[s].render_device(frames[s].data_ptr(), stream=streams[s].cuda_stream)
            if count_rays:
                total += devs[s].last_ray_counts()[:5].astype(np.int64)  # waits; also raises if a ray list overflowed
        return total

    # Warm-up that also counts the batch's rays (deterministic per view).  A handle's first frames may be reported incomplete
    # - ray lists that have to grow, or stray rays (NaN components) that switch the exact-path kernel on for the handle -:
    # rt_render_device cannot redo a frame itself, so the batch is simply rendered again until every pass is whole.
    for attempt in range(8):
        try:
            rays = batch(count_rays=True)
            break
        except native.RtError as e:
            if e.code != abi.RT_ERR_OUT_OF_RANGE or attempt == 7:
                raise
    for _ in range(max(args.warmup, 1)):
        batch()
        torch.cuda.synchronize()
    for d in devs:
        d.last_ray_counts()                            # raises if a ray list overflowed
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch()
    barrier()
    elapsed = time.perf_counter() - t0
    for d in devs:
        d.last_ray_counts()

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    r = torch.tensor(rays, dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    rays = r.cpu().numpy()

    if rank == 0:
        out = {"metric": "Mrays/s (primary+secondary), batch of %d views of reflective_bunnies at %dx%d" % (args.views, args.width, args.height),
               "value": int(rays[0]) * args.steps / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "orbit_views_%d" % args.views, "views": args.views, "width": args.width, "height": args.height,
                          "max_depth": 10, "rays_per_batch": int(rays[0]), "ms_per_view": elapsed / args.steps / max(len(mine), 1) * 1e3,
                          "views_per_pass": per_pass, "passes_in_flight": depth,
                          "parallelism": "view v on rank v %% %d, no collective" % world}}
        if args.cpu_sample_views > 0 and world == 1:
            from oracle import orc
            s = orc.OracleScene.from_file(scene_path, model_dir)
            s.resize(args.width, args.height)
            cores = min(len(os.sched_getaffinity(0)), 16)
            sample = list(range(0, args.views, max(args.views // args.cpu_sample_views, 1)))[:args.cpu_sample_views]
            cpu_rays, cpu_s = 0, 0.0
            for v in sample:
                m, w, h = cameras[mine.index(v)]
                s.set_camera_matrix(m, w, h)
                c0 = time.perf_counter()
                res = s.render(10, threads=cores, aux=False)
                cpu_s += time.perf_counter() - c0
                cpu_rays += int(res["counters"][0])
                dev.set_camera(m, w, h)
                dev.render_device(frames[0].data_ptr(), stream=handle)
                assert int(dev.last_ray_counts()[0]) == int(res["counters"][0]), "ray count of view %d differs from the oracle" % v
            out["cpu_baseline"] = {"value": cpu_rays / cpu_s / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
                                   "sample": "views %s of the batch, C++ oracle, %d threads" % (sample, cores)}
        print(json.dumps(out))
    for d in devs:
        d.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
