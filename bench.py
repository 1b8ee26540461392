#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + secondary) on reflective_bunnies at 1920x1080.

One step = one frame of the workload.  With N ranks (one per GPU) the frame is tile-sharded: rank r
renders the 32x32 tiles t with t % N == r of the same frame, the shard buffers are all-gathered over
RCCL and every rank assembles the frame - the total work is fixed, so scaling is "strong".

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

The timed frames are rendered the way a renderer with a queue of frames would: --views-per-pass frames travel through
the recursion depths together (rt_scene_set_views; by default up to eight 1080p frames' worth of pixels per rank), and a
rank keeps --frames-in-flight (default 3) such passes queued, each on its own scene handle and stream.  `value` is
that rate; the "sequential" object of the JSON line is the rate with one frame at a time.

Rank 0 prints one JSON line.  The scene's bunny.obj is not part of the reference repository; a
deterministic stand-in of 81 920 triangles is generated (rust_raytracer_amd/meshes.py) unless a real
bunny.obj is found in --model-dir.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: scene file, width, height, max_depth
    "reflective_bunnies_1080p": ("reflective_bunnies.json", 1920, 1080, 10),
    # the same with the bunny replaced by a sphere of the size and triangle count of the reference's res/models/sphere_high.obj
    # (an icosphere of radius 0.5, 20 480 triangles): a second stand-in for the bunny.obj the reference does not ship
    "reflective_bunnies_1080p_sphere_standin": ("reflective_bunnies.json", 1920, 1080, 10),
    "cornell_box_720p": ("cornell_box.json", 1280, 720, 10),
    "mirror_sphere_1080p_depth16": ("mirror_sphere.json", 1920, 1080, 16),
    "infinity_santa_1080p_depth16": ("infinity_santa.json", 1920, 1080, 16),
    # synthetic many-mesh scenes (rust_raytracer_amd/synthetic.py): grid x grid instances over `meshes` distinct
    # jittered icospheres of 81 920 triangles each - a working set far beyond the caches
    "instance_grid_2p6m_4k": ("synthetic:grid=8,meshes=32,level=6", 3840, 2160, 10),
    "instance_grid_10m_4k": ("synthetic:grid=16,meshes=128,level=6", 3840, 2160, 10),
}

# Algorithmic work of the reference's un-pruned traversal (what intersect.rs would execute), per
# unit of the metric (one ray = one Scene::intersect call), counted by the oracle:
#   bytes = 32*aabb_tests + 36*mesh_triangle_tests + 48*mesh_hits + 256*instance_transforms + 3*W*H
#   flops = 27*aabb_tests + 57*(mesh_triangle_tests + triangle_tests) + 40*sphere_tests + 14*plane_tests
#           + 90*instance_transforms + 60*light_shades
# The BVH walk kernel (rt_traverse) owns the aabb_tests and mesh_triangle_tests terms; the instance
# transforms and analytic primitives belong to rt_setup, normals and light sums to rt_shade.
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E
FP32_NO_FMA_PEAK_TFLOPS = 78.6   # 157.3 TFLOP/s vector FP32 counts an FMA as two; contraction is forbidden here


def algorithmic_work(counters, width, height):
    from rust_raytracer_amd import abi
    c = {n: int(v) for n, v in zip(abi.COUNTER_NAMES, counters)}
    walk_bytes = 32 * c["aabb_tests"] + 36 * c["mesh_triangle_tests"]
    walk_flops = 27 * c["aabb_tests"] + 57 * c["mesh_triangle_tests"]
    nbytes = walk_bytes + 48 * c["mesh_hits"] + 256 * c["instance_transforms"] + 3 * width * height
    flops = (walk_flops + 57 * c["triangle_tests"] + 40 * c["sphere_tests"] + 14 * c["plane_tests"]
             + 90 * c["instance_transforms"] + 60 * c["light_shades"])
    return {"bytes": nbytes, "flops": flops, "walk_bytes": walk_bytes, "walk_flops": walk_flops}


def cpu_baseline(scene_path, model_dir, width, height, max_depth, budget_s=15.0):
    """The CPU oracle (C++ restatement of the reference renderer, row-parallel like native_app's
    rayon loop) timed on this host; also returns its un-pruned traversal statistics."""
    from oracle import orc
    s = orc.OracleScene.from_file(scene_path, model_dir)
    s.resize(width, height)
    cores = min(len(os.sched_getaffinity(0)), 16)
    t0 = time.perf_counter()
    r = s.render(max_depth, threads=cores, aux=False)
    dt = time.perf_counter() - t0
    frames, total = 1, dt
    best = dt
    while total + best < budget_s and frames < 32:
        t0 = time.perf_counter()
        s.render(max_depth, threads=cores, aux=False)
        d = time.perf_counter() - t0
        best = min(best, d)
        total += d
        frames += 1
    rays = int(r["counters"][0])
    return {"value": rays / best / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d full %dx%d frames of the workload (%.2f s best, %d rays each), C++ oracle, %d threads"
                      % (frames, width, height, best, rays, cores)}, r["counters"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="reflective_bunnies_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--model-dir", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="independent frames a rank keeps queued (own scene handle, stream and buffers each); 1 = one after the other")
    ap.add_argument("--views-per-pass", type=int, default=0,
                    help="frames that travel through the recursion depths together (rt_scene_set_views); 0 = chosen from --steps and --gpus")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="all ranks share cuda:0 and gather over gloo: checks the N>1 code path on a one-GPU box, not a measurement")
    ap.add_argument("--watchdog", type=int, default=900, help="seconds after which a stuck run prints its stacks and exits (0 = off)")
    args = ap.parse_args()
    if args.watchdog:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)

    import numpy as np
    import torch
    import torch.distributed as dist
    from rust_raytracer_amd import abi, meshes, native, sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus %d needs a launch with torch.distributed.run (one rank per GPU)" % args.gpus)
        sys.exit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 and args.rehearse_on_one_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    scene_file, width, height, max_depth = WORKLOADS[args.workload]
    model_dir = args.model_dir
    if scene_file.startswith("synthetic:"):
        from rust_raytracer_amd import synthetic
        opts = dict(kv.split("=") for kv in scene_file[len("synthetic:"):].split(","))
        if model_dir is None:
            model_dir = os.path.join("/tmp", "rust_raytracer_amd_grid_rank%d_%d" % (rank, os.getuid()))
        os.makedirs(model_dir, exist_ok=True)
        scene_path = os.path.join(model_dir, args.workload + ".json")
        with open(scene_path, "w") as fh:
            fh.write(synthetic.instance_grid_scene(model_dir, int(opts["grid"]), int(opts["meshes"]), int(opts["level"]), width, height))
    else:
        scene_path = os.path.join(ROOT, "scenes", scene_file)
        if model_dir is None and args.workload.endswith("_sphere_standin"):
            model_dir = os.path.join("/tmp", "rust_raytracer_amd_sphere_models_rank%d_%d" % (rank, os.getuid()))
            os.makedirs(model_dir, exist_ok=True)
            with open(os.path.join(model_dir, "bunny.obj"), "w") as fh:
                fh.write(meshes.obj_text(*meshes.uv_sphere(5)))
        elif model_dir is None:
            model_dir = meshes.ensure_model_dir(os.path.join("/tmp", "rust_raytracer_amd_models_rank%d_%d" % (rank, os.getuid())))

    host = native.HostScene.load(scene_path, model_dir)
    host.resize_screen(width, height)
    m, w, h = host.camera()
    n_tris = host.desc().n_mesh_triangles

    # A frame is a chain of short dependent launches (five per recursion depth), so one frame alone leaves the
    # chip partly idle at every link.  A rank therefore keeps `depth` frames queued, as a renderer producing an
    # animation or a batch of views does: slot k has its own scene handle (own ray lists), stream and output buffers,
    # frame i goes to slot i % depth.  Every frame is still rendered completely, gathered and assembled.
    depth = max(1, args.frames_in_flight)
    # Frames of one pass share every launch (rt_scene_set_views): the fixed cost of a recursion depth - launches, prefix
    # tables, the longest BVH walk - is paid once for all of them, which matters most for the small shards of many ranks.
    views = args.views_per_pass
    if views <= 0:
        # as many as make eight 1080p frames' worth of pixels per rank (4: 9.7, 8: 10.4, 12: 10.3 Grays/s on one GPU),
        # leaving at least two passes to overlap
        cap = max(1, min(abi.RT_MAX_VIEWS, 8 * 1920 * 1080 * world // (w * h)))
        views = max([v for v in range(1, cap + 1) if args.steps // v >= 2] or [1])
    views = max(1, min(views, abi.RT_MAX_VIEWS, max(args.steps, 1)))
    passes, leftover = divmod(args.steps, views)       # the frames that do not fill a pass get a smaller pass of their own

    def host_staged_gather(gathered, shard):
        # Rehearsal only.  gloo's own device path puts high-priority streams behind events, and with two processes
        # on one GPU such a wait can hang; go through host memory instead (this stops the pipeline at every frame).
        torch.cuda.current_stream().synchronize()
        mine = shard.cpu()
        everyone = torch.empty(mine.numel() * world, dtype=torch.uint8)
        dist.all_gather_into_tensor(everyone, mine)
        gathered.copy_(everyone)

    class Slot:
        """One pass in flight: a scene handle (own ray lists), a stream and the buffers of `views` frames."""

        def __init__(self, views, dev=None, stream=None):
            self.views = views
            self.dev = dev or native.DeviceScene(host.desc(), local_rank)
            self.dev.set_views(np.tile(np.asarray(m, dtype=np.float32).reshape(1, 16), (views, 1)), w, h)
            self.stream = stream or torch.cuda.Stream()
            handle = self.stream.cuda_stream
            dev = self.dev

            def render_shard(out):
                dev.render_device(out.data_ptr(), max_depth=max_depth, shard_index=rank, shard_count=world, stream=handle)

            def assemble(gathered, frame):
                dev.assemble_shards(gathered.data_ptr(), frame.data_ptr(), world, stream=handle)

            self.sf = sharding.ShardedFrame(w, h, 3, rank, world, "cuda", render_shard, assemble,
                                            host_staged_gather if args.rehearse_on_one_gpu else None, views=views)

        def step(self):
            with torch.cuda.stream(self.stream):       # the all-gather orders itself against torch's current stream
                self.sf.step()

    # HIP gives a process four hardware queues and torch's pool streams take them in turn (stream i of the pool uses
    # queue i mod 4): passes only overlap when their streams sit on different queues (measured, 1/8 shards, three
    # passes in flight: 0.70 ms per frame on three queues, 0.87 with two passes on one queue, 1.05 with all on one).
    # So the streams are taken from the pool back to back, before anything else asks for one (with three of them the
    # fourth queue is left for the communicator's stream), and the pass of leftover frames shares a stream.
    streams = [torch.cuda.Stream() for _ in range(depth)]
    slots = [Slot(views, stream=streams[k]) for k in range(depth)]
    last_slot = Slot(leftover, stream=streams[passes % depth]) if leftover else None
    every_slot = slots + ([last_slot] if last_slot else [])
    dev, sf = slots[0].dev, slots[0].sf

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, not warm-up: the first pass on a handle allocates its ray lists, the first gather creates the
    # communicator, and a pass whose ray lists overflow is only noticed by the next one on that handle.
    for s in every_slot:
        for _ in range(2):
            s.step()
            torch.cuda.synchronize()
    for k in range(-(-args.warmup // views)):          # the warm-up frames, in passes like the timed ones
        slots[k % depth].step()
    barrier()
    t0 = time.perf_counter()
    for k in range(passes):           # args.steps frames: `views` per pass ...
        slots[k % depth].step()
    if last_slot:
        last_slot.step()              # ... and the rest
    barrier()
    elapsed = time.perf_counter() - t0
    for s in every_slot:
        s.dev.last_ray_counts()       # raises if a frame of the timed region overflowed its ray lists
    counts = dev.last_ray_counts().astype(np.int64) // views            # per frame (the views are the same frame)
    frame_check = sf.frame[:h * w * 3].cpu().numpy().reshape(h, w, 3)
    for other in every_slot:
        assert torch.equal(other.sf.frame.view(other.views, -1), sf.frame[:h * w * 3].expand(other.views, -1)), \
            "frames of the timed region differ"

    # The same number of frames one at a time on one handle and one stream, for the latency of a single frame; the
    # per-kernel pass further down runs this way too, so that a launch's duration is its own.
    one = Slot(1, dev=dev, stream=slots[0].stream)       # slot 0's handle, from here on with a single view
    one.step(); one.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one.step()
    barrier()
    elapsed_sequential = time.perf_counter() - t0
    trace_ms = dev.last_trace_ms()
    frame_ms = dev.last_render_ms()
    assert np.array_equal(one.sf.frame.cpu().numpy().reshape(h, w, 3), frame_check), "a frame rendered alone differs"

    # Per-kernel durations, live: the same steps once more with every launch of the four per-depth
    # kernels bracketed by HIP events on the launch stream (outside the timed region above, so that
    # the event records do not touch `value`).
    kernel_ms = {k: [0.0, 0] for k in dev.LEVEL_KERNELS}
    dev.set_kernel_timing(True)
    for _ in range(args.steps):
        one.step()
        for k, (ms, n) in dev.last_kernel_ms().items():        # waits for this frame's last event
            kernel_ms[k][0] += ms
            kernel_ms[k][1] += n
    dev.set_kernel_timing(False)
    barrier()

    t = torch.tensor([elapsed, elapsed_sequential], dtype=torch.float64, device="cuda")
    rays = torch.tensor(counts[:5], dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed, elapsed_sequential = (float(x) for x in t.cpu())
    rays = rays.cpu().numpy()

    if rank == 0:
        rays_per_frame = int(rays[0])
        value = rays_per_frame * args.steps / elapsed / 1e6
        out = {
            "metric": "Mrays/s (primary+secondary) at 1080p reflective_bunnies" if args.workload == "reflective_bunnies_1080p"
                      else "Mrays/s (primary+secondary) " + args.workload,
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "scene": scene_file, "width": w, "height": h, "max_depth": max_depth,
                       "mesh": ("%d mesh triangles, procedural" % n_tris if scene_file.startswith("synthetic:") else
                                "stand-in meshes (procedural, %d triangles) unless --model-dir holds the real ones" % n_tris)
                               if n_tris else "none",
                       "rays_per_frame": rays_per_frame, "primary": int(rays[1]), "shadow": int(rays[2]),
                       "reflected": int(rays[3]), "transmitted": int(rays[4]),
                       "frames_in_flight": depth * views, "passes_in_flight": depth, "views_per_pass": views,
                       "parallelism": ("tile-sharded x%d, scene replicated, %s" % (world, "REHEARSAL: ranks share one GPU, gloo"
                                       if args.rehearse_on_one_gpu else "RCCL all-gather")) if world > 1 else "single GPU"},
            # one frame at a time (the next is issued behind the previous one on the same stream)
            "sequential": {"value": rays_per_frame * args.steps / elapsed_sequential / 1e6, "unit": "Mrays/s",
                           "ms_per_step": elapsed_sequential / args.steps * 1e3},
        }
        baseline = None
        ref_counters = None
        if not args.no_cpu_baseline and world == 1:
            baseline, ref_counters = cpu_baseline(scene_path, model_dir, w, h, max_depth)
            out["cpu_baseline"] = baseline
            # the frame the GPU produced must agree with the oracle's ray statistics
            assert [int(x) for x in ref_counters[:5]] == [int(x) for x in rays[:5]], "ray counts differ from the oracle"
        # the dominant kernel is the BVH walk; its average launch duration comes from the events above
        walk_ms, walk_launches = kernel_ms["rt_traverse"]
        per_frame_launches = walk_launches // args.steps if args.steps else 0
        avg_launch_ms = walk_ms / walk_launches if walk_launches else None
        roofline = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                    "kernel": "rt_traverse", "launches_per_frame": per_frame_launches, "avg_launch_ms": avg_launch_ms,
                    "kernels_ms_per_frame": {k: (v[0] / args.steps if args.steps else None) for k, v in kernel_ms.items()},
                    "trace_ms_per_frame": trace_ms, "frame_kernel_ms": frame_ms}
        if ref_counters is not None and walk_launches:
            work = algorithmic_work(ref_counters, w, h)
            # per launch: the walk's share of the per-ray figure x the rays one launch handles (a frame's rays
            # spread over its launches)
            roofline["achieved"] = work["walk_bytes"] / per_frame_launches / (avg_launch_ms * 1e-3) / 1e9
            roofline["frac"] = roofline["achieved"] / HBM_PEAK_GBS
            roofline["algorithmic_bytes_per_ray"] = work["walk_bytes"] / rays_per_frame
            roofline["algorithmic_flops_per_ray"] = work["walk_flops"] / rays_per_frame
            roofline["rays_per_launch"] = rays_per_frame / per_frame_launches
            walk_tflops = work["walk_flops"] / per_frame_launches / (avg_launch_ms * 1e-3) / 1e12
            roofline["valu"] = {"achieved": walk_tflops, "peak": FP32_NO_FMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": walk_tflops / FP32_NO_FMA_PEAK_TFLOPS}
            # all kernels of the recursion depths against the whole per-ray figure
            roofline["whole_trace"] = {"algorithmic_bytes_per_ray": work["bytes"] / rays_per_frame,
                                       "algorithmic_flops_per_ray": work["flops"] / rays_per_frame,
                                       "achieved_GBs": work["bytes"] / (trace_ms * 1e-3) / 1e9,
                                       "achieved_TFLOPs": work["flops"] / (trace_ms * 1e-3) / 1e12}
        traffic_file = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(traffic_file):
            roofline["traffic"] = json.load(open(traffic_file)).get("hbm_bytes_per_launch")
        out["roofline"] = roofline
        out["frame_checksum"] = int(np.asarray(frame_check, dtype=np.uint64).sum())
        print(json.dumps(out))
    for s in every_slot:
        s.dev.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
