#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + secondary) on reflective_bunnies at 1920x1080.

One step = one frame of the workload.  With N ranks (one per GPU) every frame is tile-sharded: rank r
renders the 32x32 tiles t with t % N == r, the shard buffers are all-gathered over RCCL and every rank
assembles the frame - the total work is fixed, so scaling is "strong".

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

The timed frames are rendered the way a renderer with a queue of frames would: --views-per-pass frames travel through
the recursion depths together (rt_scene_set_views; by default up to eight 1080p frames' worth of pixels per rank), and a
rank keeps --frames-in-flight (default 3) such passes queued, each on its own scene handle and stream.  Every frame in
flight has its own camera: the scene's camera turned a little further from frame to frame (a camera drag), so no two
frames of a pass trace the same rays; `value` counts the rays of the frames actually rendered.  The timed region of
exactly --steps frames is repeated --repeats times and the median is reported.  The "sequential" object of the JSON
line is the rate with one frame at a time at the scene's own camera.

`roofline` describes the mode `value` is timed in: the per-kernel durations come from passes of --views-per-pass frames
(HIP events around every launch, one pass at a time, outside the timed region), the kernel with the largest share there
is the one reported, the other per-depth kernel stands beside it (`roofline.other`), and what the same kernels take with
one frame at a time is kept under `roofline.one_frame_at_a_time`.

--exercise-collective (with --gpus 1): the single rank initialises the RCCL process group and sends every pass through
the real all_gather_into_tensor and the assemble step, so that the N > 1 branch runs on a one-GPU box (its bytes checked
against frames rendered without it); not a measurement of scaling.

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

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E
# 157.3 TFLOP/s vector FP32 counts an FMA as two operations; contraction is forbidden here (bit-exact hit decisions), so the
# usable vector peak is one operation per lane and issue slot
FP32_NO_FMA_PEAK_TFLOPS = 78.6
# Operations of the two tests of the BVH walk, counted as the reference's arithmetic spells them out:
#   box (AABB::intersect, intersect.rs:185-226): 6 sub + 6 mul + 12 min/max + 3 compares               = 27
#   triangle (Triangle::intersect, intersect.rs:141-166): 2 cross (18) + 4 dot (20) + 2 sub (6) + 1 div + 3 mul + compares = 57
FLOPS_PER_BOX_TEST = 27
FLOPS_PER_TRIANGLE_TEST = 57
BOXES_PER_NODE_VISIT = 4         # the device walk tests the four entries of a 4-wide node at every visit


def reference_walk_work(counters):
    """Work of the reference's own un-pruned walk (what intersect.rs would execute for this frame), counted by the
    oracle: not what the device does - its walk is pruned - but the yardstick a CPU renderer is held to."""
    from rust_raytracer_amd import abi
    c = {n: int(v) for n, v in zip(abi.COUNTER_NAMES, counters)}
    return {"bytes": 32 * c["aabb_tests"] + 36 * c["mesh_triangle_tests"],
            "flops": FLOPS_PER_BOX_TEST * c["aabb_tests"] + FLOPS_PER_TRIANGLE_TEST * c["mesh_triangle_tests"]}


def list_kernel_bytes(counts, n_lights, setup_launches, levels):
    """Bytes the two list kernels have to move per frame (ray records 32 B, candidate keys 8 B, hit records 48 B, colour
    sums 24 B), from the frame's ray counts: what each record costs once written and once read.  Shadow rays are not
    stored: setting them up reads the 48-byte hit record once per light.  When rt_setup only ran at depth 0
    (setup_launches < levels), rt_shade did the setting up of every later depth and is charged with it."""
    rays, primary, shadow, reflected, transmitted = (int(x) for x in counts[:5])
    listed = primary + reflected + transmitted          # rays of the segmented lists; the rest are shadow rays
    hits = shadow // max(n_lights, 1)
    setup_first = primary * (32 + 8)                                                # depth 0: every ray read, its key written
    setup_later = (reflected + transmitted) * (32 + 8) + shadow * (48 + 8)
    shade = listed * (32 + 8) + hits * 48 + (reflected + transmitted) * 32 + listed * 24 \
        + hits * (48 + 8 * n_lights + 24)               # ... and the Phong pass: hit record, shadow keys, colour sums
    if setup_launches < levels:
        return {"rt_setup": setup_first, "rt_shade": shade + setup_later}
    return {"rt_setup": setup_first + setup_later, "rt_shade": shade}


def cpu_baseline(scene_path, model_dir, width, height, max_depth, budget_s=15.0):
    """The CPU oracle (C++ restatement of the reference renderer, row-parallel like native_app's
    rayon loop) timed on this host with every core the process may use; also returns its un-pruned traversal
    statistics."""
    from oracle import orc
    s = orc.OracleScene.from_file(scene_path, model_dir)
    s.resize(width, height)
    cores = len(os.sched_getaffinity(0))
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
            "host_cpus": os.cpu_count(),
            "sample": "%d full %dx%d frames of the workload at the scene's own camera (%.2f s best, %d rays each), C++ oracle, "
                      "%d threads = every core this process may run on (os.cpu_count() of the host: %d)"
                      % (frames, width, height, best, rays, cores, os.cpu_count())}, r["counters"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0,
                    help="frames in the timed region; default 96 per rank up to four ranks (96, 192, 384, 384 for 1, 2, 4, 8 ranks): three passes of "
                         "as many views as make 32 whole frames' worth of pixels per rank")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=15, help="times the timed region of --steps frames is run; the median is reported")
    ap.add_argument("--workload", default="reflective_bunnies_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--model-dir", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="independent passes a rank keeps queued (own scene handle, stream and buffers each); 1 = one after the other")
    ap.add_argument("--views-per-pass", type=int, default=0,
                    help="frames that travel through the recursion depths together (rt_scene_set_views); 0 = chosen from --steps and --gpus")
    ap.add_argument("--same-camera", action="store_true", help="every frame in flight uses the scene's own camera (as an earlier version of this benchmark did)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="all ranks share cuda:0 and gather over gloo: checks the N>1 code path on a one-GPU box, not a measurement")
    ap.add_argument("--exercise-collective", action="store_true",
                    help="with --gpus 1: initialise the RCCL process group (world size 1) and send every pass through all_gather_into_tensor + assemble")
    ap.add_argument("--watchdog", type=int, default=900, help="seconds after which a stuck run prints its stacks and exits non-zero (0 = off)")
    args = ap.parse_args()
    # stdout is for the one JSON line: whatever libraries print on the way (RCCL's version banner when a communicator is
    # created) goes to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if args.watchdog:
        # a gather that never completes must end the run with a failure, not hang the box: dump the stacks and _exit(1)
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)

    import numpy as np
    import torch
    import torch.distributed as dist
    from rust_raytracer_amd import abi, meshes, native, sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.steps <= 0:
        # (a rehearsal's ranks share one GPU's memory: it keeps to 96)
        args.steps = 96 if args.rehearse_on_one_gpu else 96 * min(max(world, 1), 4)
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
    elif args.exercise_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    collective = world > 1 or args.exercise_collective

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
    n_lights = host.desc().n_lights

    # A frame is a chain of short dependent launches (four per recursion depth), so one frame alone leaves the
    # chip partly idle at every link.  A rank therefore keeps `depth` passes queued, as a renderer producing an
    # animation or a batch of views does: slot k has its own scene handle (own ray lists), stream and output buffers,
    # pass i goes to slot i % depth.  Every frame is still rendered completely, gathered and assembled.
    depth = max(1, args.frames_in_flight)
    # Frames of one pass share every launch (rt_scene_set_views): the fixed cost of a recursion depth - launches, prefix
    # tables, the longest BVH walk - is paid once for all of them, which matters most for the small shards of many ranks.
    views = args.views_per_pass
    if views <= 0:
        # as many as make thirty-two 1080p frames' worth of pixels per rank (measured on the headline frame: 16 views per pass
        # 15.9 Grays/s, 24: 16.2, 32: 16.2; the library takes up to 2^28 pixels per pass), leaving at least two passes to overlap
        cap = max(1, min(abi.RT_MAX_VIEWS, 32 * 1920 * 1080 * world // (w * h)))
        views = max([v for v in range(1, cap + 1) if args.steps // v >= 2] or [1])
    views = max(1, min(views, abi.RT_MAX_VIEWS, max(args.steps, 1)))
    passes, leftover = divmod(args.steps, views)       # the frames that do not fill a pass get a smaller pass of their own

    # One camera per frame in flight: camera 0 is the scene's own, camera k the same turned by a drag of k * 12 pixels
    # (Scene::turn_camera, scene.rs:76-97), alternately to the left and to the right.
    n_cameras = depth * views + leftover
    position0, orientation0 = host.camera_pose()
    cameras = [np.asarray(m, dtype=np.float32).reshape(16)]
    for k in range(1, n_cameras):
        if args.same_camera:
            cameras.append(cameras[0])
            continue
        host.set_camera_pose(position0, orientation0)
        step = 12.0 * ((k % 96 + 1) // 2)             # (from the 97th frame on the cameras repeat)
        host.turn_camera(0.0, 0.0, step if k % 2 else -step, 0.0)
        cameras.append(np.asarray(host.camera()[0], dtype=np.float32).reshape(16))
    host.set_camera_pose(position0, orientation0)

    def host_staged_gather(gathered, shard):
        # Rehearsal only.  gloo's own device path puts high-priority streams behind events, and with two processes
        # on one GPU such a wait can hang; go through host memory instead (this stops the pipeline at every frame).
        torch.cuda.current_stream().synchronize()
        mine = shard.cpu()
        everyone = torch.empty(mine.numel() * world, dtype=torch.uint8)
        dist.all_gather_into_tensor(everyone, mine)
        gathered.copy_(everyone)

    class Slot:
        """One pass in flight: a scene handle (own ray lists), a stream and the buffers of its frames, one per camera."""

        def __init__(self, cams, dev=None, stream=None, gather=True):
            self.views = len(cams)
            self.cams = cams
            self.dev = dev or native.DeviceScene(host.desc(), local_rank)
            self.dev.set_views(np.stack(cams), w, h)
            self.stream = stream or torch.cuda.Stream()
            handle = self.stream.cuda_stream
            dev = self.dev

            def render_shard(out):
                dev.render_device(out.data_ptr(), max_depth=max_depth, shard_index=rank, shard_count=world, stream=handle)

            def assemble(gathered, frame):
                dev.assemble_shards(gathered.data_ptr(), frame.data_ptr(), world, stream=handle)

            self.sf = sharding.ShardedFrame(w, h, 3, rank, world, "cuda", render_shard, assemble,
                                            host_staged_gather if args.rehearse_on_one_gpu else None, views=self.views,
                                            always_gather=args.exercise_collective and world == 1 and gather)

        def step(self):
            with torch.cuda.stream(self.stream):       # the all-gather orders itself against torch's current stream
                self.sf.step()

        def frames(self):
            return self.sf.frame.view(self.views, h, w, 3)

    # HIP gives a process four hardware queues and torch's pool streams take them in turn (stream i of the pool uses
    # queue i mod 4): passes only overlap when their streams sit on different queues (measured, 1/8 shards, three
    # passes in flight: 0.70 ms per frame on three queues, 0.87 with two passes on one queue, 1.05 with all on one).
    # So the streams are taken from the pool back to back, before anything else asks for one (with three of them the
    # fourth queue is left for the communicator's stream), and the pass of leftover frames shares a stream.
    streams = [torch.cuda.Stream() for _ in range(depth)]
    slots = [Slot(cameras[k * views:(k + 1) * views], stream=streams[k]) for k in range(depth)]
    last_slot = Slot(cameras[depth * views:], stream=streams[passes % depth]) if leftover else None
    every_slot = slots + ([last_slot] if last_slot else [])
    dev = slots[0].dev

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, not warm-up.  A handle sizes its work lists by what its frames have held (rt_last_list_use): one frame of the
    # pass's first camera, counted, and the lists of the whole pass are made the right length at once - instead of four
    # rays per pixel and depth first, which for three passes of 32 views would be most of the device's memory.
    probe = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda")
    for s in every_slot:
        s.dev.set_views(np.stack(s.cams[:1]), w, h)
        for attempt in range(4):
            s.dev.render_device(probe.data_ptr(), max_depth=max_depth, shard_index=rank, shard_count=world, stream=s.stream.cuda_stream)
            try:
                s.dev.last_ray_counts()
                break
            except native.RtError:
                if attempt == 3:
                    raise
        s.dev.set_views(np.stack(s.cams), w, h)
    del probe
    # Then the passes themselves: the first one on a handle allocates its ray lists, the first gather creates the
    # communicator, and a pass whose ray lists overflow makes the next one on that handle allocate longer ones.
    for s in every_slot:
        for attempt in range(6):
            s.step()
            torch.cuda.synchronize()
            try:
                s.dev.last_ray_counts()                # raises while the lists are too short or the exact-path kernel was still off
                if attempt:
                    break
            except native.RtError:
                if attempt == 5:
                    raise
    for k in range(-(-args.warmup // views)):          # the warm-up frames, in passes like the timed ones
        slots[k % depth].step()
    regions = []
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        for k in range(passes):           # args.steps frames: `views` per pass ...
            slots[k % depth].step()
        if last_slot:
            last_slot.step()              # ... and the rest
        barrier()
        regions.append(time.perf_counter() - t0)
    elapsed = float(np.median(regions))
    # ray counts of every pass (raises if any frame since the set-up overflowed its ray lists: the report is sticky)
    slot_counts = [s.dev.last_ray_counts().astype(np.int64)[:5] for s in every_slot]
    # device memory behind the work lists and colour sums of one pass in flight (sized by what the set-up passes held)
    list_use = slots[0].dev.last_list_use()
    counts = sum(slot_counts[k % depth] for k in range(passes))
    if last_slot:
        counts = counts + slot_counts[-1]

    # The same number of frames one at a time on one handle and one stream, at the scene's own camera, for the latency
    # of a single frame; the per-kernel pass further down runs this way too, so that a launch's duration is its own.
    # Per-kernel durations in the mode `value` is timed in: passes of `views` frames on slot 0's handle, every launch of the
    # per-depth kernels bracketed by HIP events on the launch stream, one pass at a time (outside the timed region above, so
    # that the event records do not touch `value`).
    dev.set_kernel_timing(True)
    pass_kernel_ms = {k: [0.0, 0] for k in dev.LEVEL_KERNELS}
    timing_passes = max(2, min(8, -(-args.steps // views)))
    for _ in range(timing_passes):
        slots[0].step()
        for k, (ms, n) in dev.last_kernel_ms().items():        # waits for this pass's last event
            pass_kernel_ms[k][0] += ms
            pass_kernel_ms[k][1] += n
    dev.set_kernel_timing(False)
    pass_counts = dev.last_ray_counts().astype(np.int64)           # this rank's share of one pass (all of its views)
    barrier()

    one = Slot(cameras[:1], dev=dev, stream=slots[0].stream, gather=False)       # slot 0's handle, from here on with a single view
    one.step(); one.step()
    sequential = []
    for _ in range(3):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one.step()
        barrier()
        sequential.append(time.perf_counter() - t0)
    elapsed_sequential = float(np.median(sequential))
    trace_ms = dev.last_trace_ms()
    frame_ms = dev.last_render_ms()
    own_counts = dev.last_ray_counts().astype(np.int64)            # this rank's share of the frame at the scene's own camera
    frame_check = one.frames()[0].cpu().numpy()
    # every frame of the timed region against the same camera rendered alone, byte for byte
    checked = 0
    for s in every_slot:
        got = s.frames().cpu().numpy()
        for v in range(s.views):
            if v == 0 or v == s.views - 1 or args.same_camera:
                if np.array_equal(s.cams[v], cameras[0]):
                    alone = frame_check
                else:
                    one.dev.set_views(s.cams[v].reshape(1, 16), w, h)
                    one.step()
                    torch.cuda.synchronize()
                    alone = one.frames()[0].cpu().numpy()
                assert np.array_equal(got[v], alone), "frame %d of pass slot %d differs from the same camera rendered alone on %d pixels" % (
                    v, every_slot.index(s), int((got[v] != alone).any(axis=2).sum()))
                checked += 1
    one.dev.set_views(cameras[0].reshape(1, 16), w, h)
    if not args.same_camera and n_cameras > 1:
        assert len({s.frames()[v].cpu().numpy().tobytes() for s in every_slot for v in range(s.views)}) == min(n_cameras, 96), \
            "two different cameras of the timed region gave the same frame"

    # Per-kernel durations, live: the same steps once more with every launch of the per-depth
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
    rays = torch.tensor(np.concatenate([counts, own_counts[:5], own_counts[[abi.RT_CTR_WALK_NODE_VISITS, abi.RT_CTR_WALK_TRIANGLE_TESTS, abi.RT_CTR_SHADOW_NOT_TRACED]]]),
                        dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed, elapsed_sequential = (float(x) for x in t.cpu())
    rays = rays.cpu().numpy()
    region_rays, frame_rays = rays[:5], rays[5:10]
    # (the roofline below is that of THIS rank's launches: its own shard's rays and walk work, not the frame's)

    # ---- roofline: the per-depth kernel with the largest share of a frame in the mode `value` is timed in (passes of
    # `views` frames), the other per-depth kernel beside it, and both again for one frame at a time ----
    traffic_file = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
    traffic_table = json.load(open(traffic_file)).get("kernels", {}) if os.path.exists(traffic_file) and world == 1 else {}
    queued_traffic_file = os.path.join(ROOT, "profiles", "traffic_queued_%s.json" % args.workload)
    queued_traffic = json.load(open(queued_traffic_file)) if os.path.exists(queued_traffic_file) and world == 1 else {}

    def kernel_roofline(name, times, frames_per_launch, work_counts, hbm_bytes_per_launch, ref_work=None):
        """Roofline object of kernel `name`: `times` = {kernel: [summed ms, launches]} over launches that each handle
        `frames_per_launch` frames' worth of this rank's rays, `work_counts` = the counters of one such set of launches
        (one pass, or one frame)."""
        ms_sum, launches = times[name]
        if not launches:
            return None
        per_set = launches // max(n_sets[id(times)], 1)     # launches per pass / per frame
        seconds = ms_sum / launches * 1e-3
        r = {"kernel": name, "launches_per_pass": per_set, "frames_per_launch": frames_per_launch, "avg_launch_ms": ms_sum / launches,
             "ms_per_frame": ms_sum / n_sets[id(times)] / frames_per_launch, "traffic": hbm_bytes_per_launch}
        hbm = {"achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
               "what": "HBM-side bytes per launch from the PMC counters (profiles/traffic_*%s.json) / the launch's duration" % args.workload}
        if hbm_bytes_per_launch:
            hbm["achieved"] = hbm_bytes_per_launch / seconds / 1e9
            hbm["frac"] = hbm["achieved"] / HBM_PEAK_GBS
        rays_of_set = max(int(work_counts[0]), 1)
        if name == "rt_traverse":
            # the kernel's own counted work: the nodes its pruned walk visited (four box tests each) and the triangle tests it ran
            nodes, tris = int(work_counts[abi.RT_CTR_WALK_NODE_VISITS]), int(work_counts[abi.RT_CTR_WALK_TRIANGLE_TESTS])
            flops = FLOPS_PER_BOX_TEST * BOXES_PER_NODE_VISIT * nodes + FLOPS_PER_TRIANGLE_TEST * tris
            achieved = flops / per_set / seconds / 1e12
            r.update({"bound": "valu", "achieved": achieved, "peak": FP32_NO_FMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": achieved / FP32_NO_FMA_PEAK_TFLOPS,
                      "work": {"node_visits": nodes, "triangle_tests": tris, "flops": flops, "flops_per_ray": flops / rays_of_set,
                               "per": "pass of %d frames" % frames_per_launch if frames_per_launch > 1 else "frame",
                               "counted_by": "the kernel itself (RT_CTR_WALK_*), %d flop per box test x %d boxes per node visit, "
                                             "%d per triangle test" % (FLOPS_PER_BOX_TEST, BOXES_PER_NODE_VISIT, FLOPS_PER_TRIANGLE_TEST)},
                      "hbm": hbm})
            if ref_work is not None:
                # for comparison only: the rate the reference's un-pruned walk would need to finish in the same time
                r["reference_equivalent"] = {
                    "GBs": ref_work["bytes"] * frames_per_launch / per_set / seconds / 1e9,
                    "TFLOPs": ref_work["flops"] * frames_per_launch / per_set / seconds / 1e12,
                    "what": "bytes and flops of the reference's UN-PRUNED walk (oracle-counted, the scene's own camera) / this kernel's "
                            "time: an equivalent rate, not work done and no roofline fraction"}
        else:
            nbytes = list_kernel_bytes([int(x) for x in work_counts[:5]], n_lights, times["rt_setup"][1], times["rt_shade"][1])[name]
            achieved = nbytes / per_set / seconds / 1e9
            r.update({"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                      "work": {"bytes": nbytes, "bytes_per_ray": nbytes / rays_of_set,
                               "per": "pass of %d frames" % frames_per_launch if frames_per_launch > 1 else "frame",
                               "counted_by": "record sizes x the ray counts (list_kernel_bytes in bench.py)"},
                      "measured_hbm": hbm})
        return r

    n_sets = {id(pass_kernel_ms): timing_passes, id(kernel_ms): max(args.steps, 1)}
    # rank 0 of a single-GPU run also times the CPU oracle, whose counters give the reference's un-pruned walk work
    baseline, ref_counters = None, None
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        baseline, ref_counters = cpu_baseline(scene_path, model_dir, w, h, max_depth)
    ref_work = reference_walk_work(ref_counters) if ref_counters is not None else None

    def queued_hbm(name):
        # PMC traffic of a pass of `views` frames, collected per launch on passes of queued_traffic["views"] frames
        k = queued_traffic.get("kernels", {}).get(name)
        return k["hbm_bytes_per_launch"] / queued_traffic["views"] * views if k else None

    queued = {k: kernel_roofline(k, pass_kernel_ms, views, pass_counts, queued_hbm(k), ref_work) for k in ("rt_traverse", "rt_shade", "rt_setup")}
    alone = {k: kernel_roofline(k, kernel_ms, 1, own_counts, traffic_table.get(k, {}).get("hbm_bytes_per_launch"), ref_work)
             for k in ("rt_traverse", "rt_shade", "rt_setup")}
    queued = {k: v for k, v in queued.items() if v}
    alone = {k: v for k, v in alone.items() if v}
    dominant = max(queued, key=lambda k: queued[k]["ms_per_frame"])
    roofline = dict(queued[dominant])
    roofline["mode"] = "passes of %d frames, one pass at a time: the mode `value` is timed in (there with %d passes in flight)" % (views, depth)
    roofline["scope"] = "the whole frame" if world == 1 else "rank %d's launches: its 1/%d shard of every frame (work counted on that shard)" % (rank, world)
    roofline["kernels_ms_per_frame"] = {k: v["ms_per_frame"] for k, v in queued.items()}
    others = sorted((k for k in queued if k != dominant), key=lambda k: -queued[k]["ms_per_frame"])
    if others:
        roofline["other"] = queued[others[0]]
    alone_dominant = max(alone, key=lambda k: alone[k]["ms_per_frame"]) if alone else None
    roofline["one_frame_at_a_time"] = {"kernels_ms_per_frame": {k: v["ms_per_frame"] for k, v in alone.items()},
                                       "trace_ms_per_frame": trace_ms, "frame_kernel_ms": frame_ms,
                                       "dominant": alone.get(alone_dominant)}
    # every rank's own line of the same table (a rank's launches handle its own shard)
    mine = {"rank": rank, "kernel": roofline["kernel"], "bound": roofline["bound"], "achieved": roofline["achieved"], "unit": roofline["unit"],
            "frac": roofline["frac"], "avg_launch_ms": roofline["avg_launch_ms"], "kernels_ms_per_frame": roofline["kernels_ms_per_frame"]}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank == 0:
        rays_per_frame = int(frame_rays[0])
        value = int(region_rays[0]) / elapsed / 1e6
        out = {
            "metric": "Mrays/s (primary+secondary) at 1080p reflective_bunnies" if args.workload == "reflective_bunnies_1080p"
                      else "Mrays/s (primary+secondary) " + args.workload,
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": len(regions), "timed_region_s": elapsed, "timed_regions_s": [round(x, 6) for x in regions],
            "config": {"workload": args.workload, "scene": scene_file, "width": w, "height": h, "max_depth": max_depth,
                       "mesh": ("%d mesh triangles, procedural" % n_tris if scene_file.startswith("synthetic:") else
                                "stand-in meshes (procedural, %d triangles) unless --model-dir holds the real ones" % n_tris)
                               if n_tris else "none",
                       "cameras": "every frame of the scene's own camera" if args.same_camera else
                                  "%d distinct cameras: the scene's own and %d drags of it (12 pixels apart, both ways); %d of the "
                                  "frames compared byte for byte with the same camera rendered alone" % (min(n_cameras, 96), min(n_cameras, 96) - 1, checked),
                       "rays_in_timed_region": int(region_rays[0]), "mean_rays_per_frame": int(region_rays[0]) / max(args.steps, 1),
                       "rays_per_frame": rays_per_frame, "primary": int(frame_rays[1]), "shadow": int(frame_rays[2]),
                       "reflected": int(frame_rays[3]), "transmitted": int(frame_rays[4]),
                       # rays are counted as the reference traces them; of the shadow rays, these were settled without a trace
                       # because their verdict cannot change the picture (RT_CTR_SHADOW_NOT_TRACED in include/rt_hip.h)
                       "shadow_not_traced": int(rays[12]),
                       "rays_counted": "the rays the reference traces for these frames (checked against the oracle's count); the shadow rays in "
                                       "shadow_not_traced are among them - settled without a trace because their verdict cannot change the picture",
                       "frames_in_flight": depth * views, "passes_in_flight": depth, "views_per_pass": views,
                       "list_bytes_per_pass": list_use["bytes"], "list_bytes_per_view": list_use["bytes"] // max(views, 1),
                       "parallelism": ("tile-sharded x%d, scene replicated, %s" % (world, "REHEARSAL: ranks share one GPU, gloo"
                                       if args.rehearse_on_one_gpu else "RCCL all-gather")) if world > 1 else
                                      ("single GPU, every pass through an RCCL all-gather of world size 1" if args.exercise_collective else "single GPU")},
            # one frame at a time at the scene's own camera (the next is issued behind the previous one on the same stream)
            "sequential": {"value": rays_per_frame * args.steps / elapsed_sequential / 1e6, "unit": "Mrays/s",
                           "ms_per_step": elapsed_sequential / args.steps * 1e3},
        }
        if baseline is not None:
            out["cpu_baseline"] = baseline
            # the frame the GPU produced must agree with the oracle's ray statistics
            assert [int(x) for x in ref_counters[:5]] == [int(x) for x in frame_rays], "ray counts differ from the oracle"

        out["roofline"] = roofline
        if world > 1:
            out["roofline_per_rank"] = per_rank
        if args.exercise_collective:
            out["collective_exercised"] = "every pass of the timed region went through all_gather_into_tensor on an RCCL process group of world size %d and the assemble step" % world
        out["frame_checksum"] = int(np.asarray(frame_check, dtype=np.uint64).sum())
        sys.stdout.flush()
        os.dup2(real_stdout, 1)                 # the one line this program prints
        print(json.dumps(out), flush=True)
    for s in every_slot:
        s.dev.close()
    if world > 1:
        dist.barrier()
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
