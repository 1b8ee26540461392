"""HIP renderer (through the C ABI) against the CPU oracle on the same scene arrays.

Bar: primary hit primitive / instance ids and the shadow / reflection /
refraction flags bit-exact for every pixel; RGB8 within 1 LSB on >= 99.9 % of the pixels.  The
colour tolerance exists because the device sums throughput-weighted contributions where the
reference evaluates the recursion bottom-up (a last-bit difference before quantisation) and because
powf differs in the last place between libms.  Ray counts must agree exactly.
"""
import json
import os

import numpy as np
import pytest

from oracle import orc
from rust_raytracer_amd import abi, native

pytestmark = pytest.mark.gpu


def render_both(oscene, max_depth=10, queue_capacity=0):
    m, w, h = oscene.camera()
    dev = native.DeviceScene(oscene.desc(), 0)
    try:
        dev.set_camera(m, w, h)
        if queue_capacity:
            dev.debug_set_queue_capacity(queue_capacity)
        got = dev.render(max_depth=max_depth, aux=True)
        got["regrown"] = dev.debug_last_regrown()
        # Without the per-pixel flags the shadow rays of primary hits whose verdict cannot matter go untraced as well (with
        # them, only those of deeper hits): the picture has to be the same bytes either way.
        plain = dev.render(max_depth=max_depth)
        assert np.array_equal(plain["rgb"], got["rgb"]), "picture differs on %d pixels without the flags" % (plain["rgb"] != got["rgb"]).any(axis=-1).sum()
    finally:
        dev.close()
    want = oscene.render(max_depth=max_depth)
    return got, want


def assert_parity(got, want, min_within_1=0.999):
    assert np.array_equal(got["flags"], want["flags"]), "flags differ on %d pixels" % (got["flags"] != want["flags"]).sum()
    assert np.array_equal(got["prim_id"], want["prim_id"]), "prim ids differ on %d pixels" % (got["prim_id"] != want["prim_id"]).sum()
    assert np.array_equal(got["instance_id"], want["instance_id"])
    # Hitpoint::t of the primary hits, bit for bit (NaN where nothing is hit, and for the hits of NaN rays)
    assert np.array_equal(got["hit_t"], want["hit_t"], equal_nan=True), "primary t differs on %d pixels" % (
        ~((got["hit_t"] == want["hit_t"]) | (np.isnan(got["hit_t"]) & np.isnan(want["hit_t"])))).sum()
    assert got["counters"][:5].tolist() == want["counters"][:5].tolist()
    diff = np.abs(got["rgb"].astype(int) - want["rgb"].astype(int)).max(axis=2)
    within = (diff <= 1).mean()
    assert within >= min_within_1, "only %.5f of the pixels within 1 LSB (max diff %d)" % (within, diff.max())
    return (diff == 0).mean(), within


def load(scenes_dir, model_dir, name, width, height):
    s = orc.OracleScene.from_file(os.path.join(scenes_dir, name + ".json"), model_dir)
    s.resize(width, height)
    return s


def shoot(desc, origin, direction):
    """One ray through the device path: a 1x1 frame whose screen_to_world matrix maps the pixel to `origin` and the
    point one unit deeper to origin + direction (raytracer.rs:30-51).  Returns (prim_id, t) of Scene::intersect."""
    m = np.zeros((4, 4), dtype=np.float32)         # column-major: m[c] is column c
    m[2, :3] = direction
    m[3, :3] = origin
    m[3, 3] = 1.0
    dev = native.DeviceScene(desc, 0)
    try:
        dev.set_camera(m.reshape(16), 1, 1)
        r = dev.render(max_depth=1, aux=True)
    finally:
        dev.close()
    hit = bool(r["flags"][0, 0] & abi.RT_FLAG_HIT)
    assert np.isnan(r["hit_t"][0, 0]) != hit
    return (int(r["prim_id"][0, 0]), float(r["hit_t"][0, 0])) if hit else None


def test_reference_intersection_kats_on_the_device():
    """The reference's own known-answer tests (lib_raytracer/src/raytracing/intersect_tests.rs:8-161) through the HIP
    path: each case is a scene with the one primitive and a one-pixel frame whose ray is the test's ray; the hit
    distance comes back through RtAux::hit_t and is held to the reference's tolerance (utils/mod.rs:4-7:
    relative error <= 2 * f32::EPSILON).  tests/test_oracle_kat.py runs the same cases on the oracle."""
    import ctypes as C
    eps = float(np.finfo(np.float32).eps)

    def scene_with(**arrays):
        d = abi.RtSceneDesc()
        d.background.type = abi.RT_BACKGROUND_COLORED_DIRECTION
        keep = [(abi.RtMaterial * 1)()]
        d.n_materials, d.materials = 1, C.cast(keep[0], C.POINTER(abi.RtMaterial))
        for name, items in arrays.items():
            arr = (type(items[0]) * len(items))(*items)
            keep.append(arr)
            setattr(d, "n_" + name, len(items))
            setattr(d, name, C.cast(arr, C.POINTER(type(items[0]))))
        return d, keep

    def check(desc, origin, direction, expected, kind):
        got = shoot(desc, origin, direction)
        if expected is None:
            assert got is None, (origin, direction, got)
        else:
            assert got is not None, (origin, direction)
            assert got[0] == kind << 30 and abs(got[1] - expected) <= 2.0 * eps * max(abs(got[1]), abs(expected)), (origin, direction, got)

    f3 = lambda v: (abi.f32 * 3)(*[float(x) for x in v])
    sphere, keep1 = scene_with(spheres=[abi.RtSphere(f3((0, 0, 0)), 1.0, 0)])                          # intersect_tests.rs:8-19
    for origin, direction, expected in [((0, 0, -4100.0), (0, 0, 1), 4099.0), ((0, 0, -2.0), (0, 1, 0), None),
                                        ((0, 0, -2.0), (0, 0, 1), 1.0), ((0, 0, 0.0), (0, 0, 1), 1.0),
                                        ((-1, 0, -1.0), (0, 0, 1), 1.0), ((0, 0, -1.1), (0, 0, -1), None)]:     # :20-90
        check(sphere, origin, direction, expected, abi.RT_PRIM_SPHERE)
    tri = abi.RtTriangle()                                                                              # :96-102
    for k, v in enumerate([(-1, 1, 0), (1, 0, 0), (-1, -1, 0)]):
        tri.vertices[k] = f3(v)
    # Triangle::new's face normal: normalize(cross(c - a, b - a)), geometry/triangle.rs:43
    a, b, c = (np.array(tri.vertices[k][:], dtype=np.float32) for k in range(3))
    n = np.cross(c - a, b - a).astype(np.float32)
    tri.face_normal = f3(n / np.sqrt((n * n).sum(dtype=np.float32)))
    triangle, keep2 = scene_with(triangles=[tri])
    for direction, expected in [((0, 0, 1), 2.0), ((0, 0, -1), None), ((0, 1, 0), None)]:                  # :104-127
        check(triangle, (0, 0, -2.0), direction, expected, abi.RT_PRIM_TRIANGLE)
    plane, keep3 = scene_with(planes=[abi.RtPlane(f3((0, 0, -1)), 1.0, 0)])                              # :133-136
    for origin, direction, expected in [((0, 0, -2.0), (0, 1, 0), None), ((0, 0, -2.0), (0, 0, 1), 1.0),
                                        ((0, 0, 0.0), (0, 0, 1), None)]:                                 # :138-160
        check(plane, origin, direction, expected, abi.RT_PRIM_PLANE)


@pytest.mark.parametrize("width,height", [(320, 180), (1280, 720)])
def test_cornell_box(scenes_dir, model_dir, width, height):
    got, want = render_both(load(scenes_dir, model_dir, "cornell_box", width, height))
    assert_parity(got, want)


@pytest.mark.parametrize("name,width,height,depth", [
    ("reflective_bunnies", 480, 270, 10),
    ("bvh_comparison", 256, 256, 10),
    ("purple_marbles", 320, 180, 10),
    ("mirror_sphere", 320, 180, 16),
    ("infinity_santa", 320, 180, 16),
    ("santa_and_balls", 320, 180, 10),
])
def test_reference_scenes(scenes_dir, model_dir, name, width, height, depth):
    got, want = render_both(load(scenes_dir, model_dir, name, width, height), max_depth=depth)
    assert_parity(got, want)


def test_headline_workload_at_full_size(scenes_dir, tmp_path_factory):
    """The benchmark's own configuration: reflective_bunnies at 1920x1080, depth 10, the 81 920-triangle
    stand-in bunny - every pixel against the oracle."""
    from rust_raytracer_amd import meshes
    full_models = meshes.ensure_model_dir(str(tmp_path_factory.mktemp("full_models")), names=["bunny.obj"])
    s = load(scenes_dir, full_models, "reflective_bunnies", 1920, 1080)
    got, want = render_both(s)
    exact, within = assert_parity(got, want)
    assert got["counters"][0] == 17899018            # the ray count bench.py divides by
    assert exact >= 0.9999


def test_reference_render_bunny_free_pixels(scenes_dir):
    """The device path against the reference's own picture: scenes/reflective_bunnies.json without its meshes at
    3840x2160 against every pixel of render.png that no bunny can influence (tests/golden/reference_render_bunny_free.png,
    made by tests/golden_mask.py) - planes, the triangle, the four spheres with refraction and Fresnel weights, shadows,
    ten levels of recursion.  The matrices here come from the library's own transform code, not from nalgebra, hence
    the usual colour bar instead of byte equality."""
    from PIL import Image
    golden = np.array(Image.open(os.path.join(os.path.dirname(__file__), "golden", "reference_render_bunny_free.png")).convert("RGBA"))
    usable = golden[..., 3] == 255
    scene = json.load(open(os.path.join(scenes_dir, "reflective_bunnies.json")))
    scene["meshes"], scene["mesh_instances"] = [], []
    host = native.HostScene.parse(json.dumps(scene))
    m, w, h = host.camera()
    assert (w, h) == (3840, 2160)
    dev = native.DeviceScene(host.desc(), 0)
    try:
        dev.set_camera(m, w, h)
        got = dev.render(max_depth=10)["rgb"]
    finally:
        dev.close()
        host.close()
    all_diff = np.abs(got.astype(int) - golden[..., :3].astype(int)).max(axis=2)
    diff = all_diff[usable]
    assert usable.sum() > 2_300_000
    assert (diff <= 1).mean() >= 0.999, (diff <= 1).mean()
    assert (diff == 0).mean() >= 0.99, (diff == 0).mean()
    # the glass spheres (alpha 128: bunnies show through part of them): most of their pixels still agree
    glass = all_diff[golden[..., 3] == 128]
    assert glass.size > 500_000 and (glass <= 1).mean() >= 0.6, (glass <= 1).mean()


@pytest.mark.parametrize("name", ["mirror_sphere", "infinity_santa"])
def test_depth16_scenes_at_full_size(scenes_dir, tmp_path_factory, name):
    """The divergence and stack-depth configurations at their own size: 1920x1080, reflection depth 16, full-size stand-in
    meshes (66 M and 73 M rays) - every pixel against the oracle."""
    from rust_raytracer_amd import meshes
    full_models = meshes.ensure_model_dir(str(tmp_path_factory.mktemp("full_models16")))
    got, want = render_both(load(scenes_dir, full_models, name, 1920, 1080), max_depth=16)
    exact, within = assert_parity(got, want)
    print("%s 1920x1080 depth 16: %d rays, %.5f of the pixels identical, %.5f within 1 LSB" % (name, int(got["counters"][0]), exact, within))
    assert got["counters"][0] > 60_000_000


def test_instance_grid_2p6m_triangles_at_4k(tmp_path_factory):
    """The many-mesh configuration at benchmark size: 64 instances over 32 distinct meshes of 81 920 triangles
    (2.6 M triangles, far beyond the caches) at 3840x2160 - every pixel against the oracle."""
    from rust_raytracer_amd import synthetic
    d = str(tmp_path_factory.mktemp("grid_models_2p6m"))
    s = orc.OracleScene.from_text(synthetic.instance_grid_scene(d, 8, 32, 6, 3840, 2160), d)
    assert s.desc().n_mesh_triangles == 32 * 81920
    got, want = render_both(s)
    exact, within = assert_parity(got, want)
    print("instance grid 2.6 M triangles 3840x2160: %d rays, %.5f identical, %.5f within 1 LSB" % (int(got["counters"][0]), exact, within))
    assert (got["prim_id"] >> 30 == abi.RT_PRIM_MESH_TRIANGLE).sum() > 1_000_000


def test_instance_grid_10m_triangles_at_4k(tmp_path_factory):
    """The many-mesh configuration at its stated size: 256 instances over 128 distinct meshes of 81 920 triangles
    (10.5 M triangles; the scenes with more than 16 instances walk the range tree over instances) at 3840x2160 - every
    pixel against the oracle."""
    from rust_raytracer_amd import synthetic
    d = str(tmp_path_factory.mktemp("grid_models_10m"))
    s = orc.OracleScene.from_text(synthetic.instance_grid_scene(d, 16, 128, 6, 3840, 2160), d)
    assert s.desc().n_mesh_triangles == 128 * 81920
    got, want = render_both(s)
    exact, within = assert_parity(got, want)
    print("instance grid 10.5 M triangles 3840x2160: %d rays, %.5f identical, %.5f within 1 LSB" % (int(got["counters"][0]), exact, within))
    assert (got["prim_id"] >> 30 == abi.RT_PRIM_MESH_TRIANGLE).sum() > 1_000_000


def test_orbit_batch_of_32_views_at_1080p(scenes_dir, model_dir):
    """The batch-of-views configuration at full frame size: 32 poses on a circle around the headline scene at 1920x1080,
    handed over sixteen at a time (rt_scene_set_views), the two passes in flight together on two handles and two streams
    (rt_render_device) - every view's picture against the oracle's, the passes' ray counts against the sum of the
    oracle's; then the first sixteen once more through rt_render with the per-pixel ids, flags and t of every view."""
    import torch
    from rust_raytracer_amd import synthetic
    s = load(scenes_dir, model_dir, "reflective_bunnies", 1920, 1080)
    host = native.HostScene.load(os.path.join(scenes_dir, "reflective_bunnies.json"), model_dir)
    host.resize_screen(1920, 1080)
    mats = []
    for position, orientation in synthetic.orbit_poses(32, centre=(0.0, 0.0, -6.0)):
        host.set_camera_pose(position, orientation)
        m, w, h = host.camera()
        mats.append(np.asarray(m, dtype=np.float32).reshape(16))
    want = []
    for m in mats:
        s.set_camera_matrix(m, w, h)
        want.append(s.render())
    devs = [native.DeviceScene(host.desc(), 0) for _ in range(2)]
    try:
        streams = [torch.cuda.Stream() for _ in range(2)]
        bufs = [torch.zeros(16 * h * w * 3, dtype=torch.uint8, device="cuda") for _ in range(2)]
        for attempt in range(4):                                        # (poses that send rays onto the exact path make a handle redo its first pass)
            for k in range(2):
                devs[k].set_views(np.stack(mats[16 * k:16 * k + 16]), w, h)
                devs[k].render_device(bufs[k].data_ptr(), stream=streams[k].cuda_stream)
            torch.cuda.synchronize()
            try:
                counts = [devs[k].last_ray_counts() for k in range(2)]
                break
            except native.RtError:
                if attempt == 3:
                    raise
        for k in range(2):
            got = bufs[k].cpu().numpy().reshape(16, h, w, 3)
            for v in range(16):
                diff = np.abs(got[v].astype(int) - want[16 * k + v]["rgb"].astype(int)).max(axis=2)
                assert (diff <= 1).mean() >= 0.999, (k, v, (diff <= 1).mean())
            assert counts[k][:5].tolist() == sum(np.asarray(x["counters"][:5], dtype=np.int64) for x in want[16 * k:16 * k + 16]).tolist()
        together = devs[0].render(aux=True)
        for v in range(16):
            assert_parity({key: together[key][v] for key in ("rgb", "prim_id", "instance_id", "flags", "hit_t")} | {"counters": want[v]["counters"]}, want[v])
    finally:
        for d in devs:
            d.close()
        host.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_pruning_bound_adversarial(tmp_path_factory, seed):
    """synthetic.pruning_stress_scene: a mesh that is walked with distance pruning (|e1||e2| = 0.1225, bound 0.125) seen
    through parallel rays that run inside triangle planes and along box faces, slivers whose Moeller-Trumbore
    determinant is 1e-5 .. 2e-5 stacked behind each other, point lights exactly at candidates' distances.  Whatever
    the pruned, near-first walk skips must not change a winner, a shadow verdict or a t."""
    from rust_raytracer_amd import synthetic
    d = str(tmp_path_factory.mktemp("stress_models"))
    text, cameras = synthetic.pruning_stress_scene(d, seed)
    s = orc.OracleScene.from_text(text, d)
    dev = native.DeviceScene(s.desc(), 0)
    try:
        sliver_hits = 0
        for k, (m, w, h) in enumerate(cameras):
            m = np.asarray(m, dtype=np.float32)
            s.set_camera_matrix(m, w, h)
            dev.set_camera(m, w, h)
            got, want = dev.render(max_depth=6, aux=True), s.render(max_depth=6)
            assert_parity(got, want)
            mesh = got["prim_id"] >> 30 == abi.RT_PRIM_MESH_TRIANGLE
            assert mesh.sum() > w * h // 8
            if k == 1:
                sliver_hits = len(np.unique(got["prim_id"][mesh]))
        assert sliver_hits == 6                      # every sliver wins somewhere
    finally:
        dev.close()


@pytest.mark.parametrize("levels", [2, 64])
def test_packet_walk(scenes_dir, model_dir, tmp_path_factory, monkeypatch, levels):
    """The BVH walks of the first two depths, or of all, as packet walks (rt_traverse_packet: a wave's pairs of one instance
    walk the top of the tree together - scalar node loads, one ballot per child - and go on lane by lane from where they
    part): hits, t and colours are those of the reference on the headline scene, on an instance grid over several meshes
    and on a frame whose lists are too short at first."""
    from rust_raytracer_amd import synthetic
    monkeypatch.setenv("RT_WALK_PACKET_LEVELS", str(levels))            # read when the scene handle is created
    s = load(scenes_dir, model_dir, "reflective_bunnies", 480, 270)
    got, want = render_both(s)
    assert_parity(got, want)
    assert (got["prim_id"] >> 30 == abi.RT_PRIM_MESH_TRIANGLE).sum() > 480 * 270 // 40
    got, want = render_both(s, queue_capacity=30000)
    assert got["regrown"] >= 1
    assert_parity(got, want)
    d = str(tmp_path_factory.mktemp("packet_grid"))
    g = orc.OracleScene.from_text(synthetic.instance_grid_scene(d, 4, 4, 3, 320, 180), d)
    got, want = render_both(g)
    assert_parity(got, want)


def test_rays_from_far_away(tmp_path_factory):
    """synthetic.far_origin_scene: rays that start 2^21 units away on every axis pass whatever box of a mesh comes near them (the slab test of
    intersect.rs:185-226 cannot tell the boxes apart at that distance), and the reference tests nearly every triangle for
    each of them.  Such pairs go to traverse_heavy_pairs - a block tests all leaves side by side - up to a number per depth,
    which this frame exceeds: both ways of computing a pair must give the reference's hit, t and colour."""
    from rust_raytracer_amd import synthetic
    d = str(tmp_path_factory.mktemp("far_models"))
    text, cameras = synthetic.far_origin_scene(d, 5)
    s = orc.OracleScene.from_text(text, d)
    dev = native.DeviceScene(s.desc(), 0)
    try:
        heavy = []
        for m, w, h in cameras:
            m = np.asarray(m, dtype=np.float32)
            s.set_camera_matrix(m, w, h)
            dev.set_camera(m, w, h)
            got, want = dev.render(max_depth=3, aux=True), s.render(max_depth=3)
            assert_parity(got, want)
            assert (got["prim_id"] >> 30 == abi.RT_PRIM_MESH_TRIANGLE).sum() > w * h // 8
            heavy.append(int(got["counters"][abi.RT_CTR_WALK_HEAVY_PAIRS]))
        assert heavy[0] >= 1024 and heavy[1] > 0, heavy               # the first frame has more such pairs than are taken per depth
    finally:
        dev.close()


@pytest.mark.parametrize("grid,n_meshes,level,width,height", [(4, 4, 3, 480, 270), (7, 5, 2, 320, 180)])
def test_instance_grid(tmp_path_factory, grid, n_meshes, level, width, height):
    """Many instances over several distinct meshes, arbitrary rotations and non-uniform scales, two lights,
    reflective and matte cells (the synthetic many-mesh configuration at a small size)."""
    from rust_raytracer_amd import synthetic
    d = str(tmp_path_factory.mktemp("grid_models"))
    text = synthetic.instance_grid_scene(d, grid, n_meshes, level, width, height)
    s = orc.OracleScene.from_text(text, d)
    got, want = render_both(s)
    assert_parity(got, want)
    assert (got["prim_id"] >> 30 == abi.RT_PRIM_MESH_TRIANGLE).sum() > width * height // 20


@pytest.mark.parametrize("analytic", ["as in the scene", "none", "too many to stage"])
def test_instances_scanned_one_by_one(tmp_path_factory, analytic):
    """Scenes of at most 16 instances scan them one by one: with the culling records staged in LDS next to the analytic
    primitives (3 x 3 and 4 x 4 grids as they are), and - when there is nothing to stage them next to, or the
    primitives are too many for LDS - with the boxes read from the instance array.  16 instances is the most this
    path sees; 17 and more walk the range tree (test_instance_grid)."""
    from rust_raytracer_amd import synthetic
    d = str(tmp_path_factory.mktemp("scan_models"))
    for grid in (3, 4):
        scene = json.loads(synthetic.instance_grid_scene(d, grid, 3, 2, 240, 135))
        assert len(scene["mesh_instances"]) == grid * grid <= 16
        if analytic == "none":
            scene["planes"], scene["spheres"], scene["triangles"] = [], [], []
        elif analytic == "too many to stage":
            material = scene["materials"][0]["name"]
            scene["spheres"] = scene.get("spheres", []) + [
                {"center": {"x": -6.0 + 0.013 * k, "y": 3.5 + 0.004 * (k % 50), "z": -14.0 - 0.002 * k}, "radius": 0.02, "material": material}
                for k in range(900)]
        s = orc.OracleScene.from_text(json.dumps(scene), d)
        got, want = render_both(s, max_depth=5)
        assert_parity(got, want)
        assert (got["prim_id"] >> 30 == abi.RT_PRIM_MESH_TRIANGLE).sum() > 240 * 135 // 20


def test_orbit_views(scenes_dir, model_dir):
    """One uploaded scene, eight camera poses on a circle around it (the batch-of-views configuration at a
    small size): only the camera matrix crosses to the device between frames."""
    from rust_raytracer_amd import synthetic
    s = load(scenes_dir, model_dir, "reflective_bunnies", 320, 180)
    host = native.HostScene.load(os.path.join(scenes_dir, "reflective_bunnies.json"), model_dir)
    host.resize_screen(320, 180)
    dev = native.DeviceScene(s.desc(), 0)
    try:
        for position, orientation in synthetic.orbit_poses(8, centre=(0.0, 0.0, -6.0)):
            host.set_camera_pose(position, orientation)
            m, w, h = host.camera()
            dev.set_camera(m, w, h)
            s.set_camera_matrix(m, w, h)
            got = dev.render(aux=True)
            want = s.render()
            assert_parity(got, want)
    finally:
        dev.close()
        host.close()


@pytest.mark.parametrize("name", ["cornell_box", "reflective_bunnies"])
def test_depth_map_mode(scenes_dir, model_dir, name):
    """Raytracer::depth_map (raytracer.rs:21-28) instead of raytrace: 2^-t * 1.5 of the primary hit."""
    s = load(scenes_dir, model_dir, name, 320, 180)
    m, w, h = s.camera()
    dev = native.DeviceScene(s.desc(), 0)
    try:
        dev.set_camera(m, w, h)
        got = dev.render(depth_map=True)
    finally:
        dev.close()
    want = s.render_depth_map()
    assert got["counters"][:5].tolist() == [w * h, w * h, 0, 0, 0]
    assert want["counters"][0] == w * h
    diff = np.abs(got["rgb"].astype(int) - want["rgb"].astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999 and diff.max() <= 1        # powf is the only inexact step
    assert len(np.unique(got["rgb"])) > 5                           # a real gradient, not a constant image


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_depth_limit(scenes_dir, model_dir, depth):
    # raytracer.rs:66: a child requested at depth == max_depth is not traced, its parent takes the background
    got, want = render_both(load(scenes_dir, model_dir, "cornell_box", 320, 180), max_depth=depth)
    assert_parity(got, want)


def test_ragged_frame_size(scenes_dir, model_dir):
    # width and height that are not multiples of the 32-pixel tile or the 8x8 wave block
    got, want = render_both(load(scenes_dir, model_dir, "reflective_bunnies", 333, 117))
    assert_parity(got, want)


def test_queue_overflow_is_harmless(scenes_dir, model_dir):
    # ray lists far too short for the frame: rt_render notices, enlarges them and renders the frame again
    s = load(scenes_dir, model_dir, "cornell_box", 320, 180)
    got, want = render_both(s, queue_capacity=1000)
    assert got["regrown"] > 0
    assert_parity(got, want)


def test_overflow_of_a_device_frame_is_reported_once(scenes_dir, model_dir):
    """rt_render_device cannot redo a frame whose ray lists overflowed: rt_last_ray_counts must say so even when further
    frames were enqueued in between (the report is sticky), say it once, and the lists grow until a frame fits."""
    import torch
    s = load(scenes_dir, model_dir, "cornell_box", 320, 180)
    m, w, h = s.camera()
    dev = native.DeviceScene(s.desc(), 0)
    try:
        dev.set_camera(m, w, h)
        want = dev.render()["rgb"]
        dev.debug_set_queue_capacity(1000)
        side = torch.cuda.Stream()
        torch.cuda.set_stream(side)
        out = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda")
        dev.render_device(out.data_ptr(), stream=side.cuda_stream)      # far too short lists
        torch.cuda.synchronize()
        dev.render_device(out.data_ptr(), stream=side.cuda_stream)      # a later frame must not erase the report
        dev.render_device(out.data_ptr(), stream=side.cuda_stream)
        with pytest.raises(native.RtError) as e:
            dev.last_ray_counts()
        assert e.value.code == abi.RT_ERR_OUT_OF_RANGE
        for _ in range(20):                                              # the lists double after every overflow seen
            dev.render_device(out.data_ptr(), stream=side.cuda_stream)
            try:
                counts = dev.last_ray_counts()
                break
            except native.RtError:
                continue
        else:
            raise AssertionError("the lists never grew enough")
        assert counts[0] > w * h
        assert dev.last_ray_counts().tolist() == counts.tolist()        # reported once: asking again is fine
        assert np.array_equal(out.cpu().numpy().reshape(h, w, 3), want)
    finally:
        dev.close()


def test_lists_are_sized_by_what_the_handle_has_seen(scenes_dir, model_dir):
    """A handle's first frame of a size gets lists of four rays per pixel and depth (1.2 KB of device memory per pixel);
    once a frame has been counted the next one gets lists half as long again as the fullest depth held: 1.0 GiB for the
    1080p headline frame, the same bytes out, no overflow.  A camera close to a glass sphere (two child rays per hit) then
    needs more than was seen: the frame is reported, the next one grows, and the picture ends up the oracle's."""
    import torch
    s = load(scenes_dir, model_dir, "reflective_bunnies", 1920, 1080)
    m, w, h = s.camera()
    dev = native.DeviceScene(s.desc(), 0)
    try:
        dev.set_camera(m, w, h)
        out = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda")
        dev.render_device(out.data_ptr())
        first = dev.last_list_use()
        first_picture = out.cpu().numpy().copy()
        assert first["ray_slots"] == 4 * first_picture.size // 3 or first["ray_slots"] == 4 * ((w + 31) // 32) * ((h + 31) // 32) * 1024
        assert first["rays"] >= w * h and first["walk_items"] > 0 and first["hits"] > 0
        dev.render_device(out.data_ptr())
        counts = dev.last_ray_counts()                                   # raises if a list overflowed
        second = dev.last_list_use()
        assert second["bytes"] < 1100 << 20 and second["bytes"] * 2 < first["bytes"], (first, second)
        for slots, used in (("ray_slots", "rays"), ("walk_slots", "walk_items"), ("hit_slots", "hits")):
            assert second[used] == first[used] and second[used] < second[slots] <= 2 * second[used] + 8192, (slots, first, second)
        assert np.array_equal(out.cpu().numpy(), first_picture)
        dev.render_device(out.data_ptr())                                # and the buffers stay as they are now
        assert dev.last_list_use()["bytes"] == second["bytes"]
        # a pose that makes more rays per pixel than the handle has seen
        host = native.HostScene.load(os.path.join(scenes_dir, "reflective_bunnies.json"), model_dir)
        try:
            host.resize_screen(w, h)
            host.set_camera_pose((3.0, -1.0, -2.8), (0.0, 0.0, 0.0))
            m2, _, _ = host.camera()
        finally:
            host.close()
        m2 = np.asarray(m2, dtype=np.float32).reshape(16)
        s.set_camera_matrix(m2, w, h)
        want = s.render()
        dev.set_camera(m2, w, h)
        for attempt in range(6):
            dev.render_device(out.data_ptr())
            try:
                got_counts = dev.last_ray_counts()
                break
            except native.RtError as e:
                assert e.code == abi.RT_ERR_OUT_OF_RANGE
        else:
            raise AssertionError("the lists never grew enough")
        assert got_counts[:5].tolist() == [int(x) for x in want["counters"][:5]]
        diff = np.abs(out.cpu().numpy().reshape(h, w, 3).astype(int) - want["rgb"].astype(int)).max(axis=2)
        assert (diff <= 1).mean() >= 0.999
        assert counts[0] > w * h
    finally:
        dev.close()


def test_a_hundred_views_in_one_pass(scenes_dir, model_dir):
    """A pass takes up to RT_MAX_VIEWS = 256 views; rt_generate takes 32 at a time, so a pass of 100 is four launches that
    fill the same list segments one after the other.  Every view's bytes equal the frame rendered alone, the pass's ray counts
    their sum - whole frames and as one of three tile shards."""
    import torch
    from rust_raytracer_amd import synthetic
    w, h, n = 320, 180, 100
    host = native.HostScene.load(os.path.join(scenes_dir, "reflective_bunnies.json"), model_dir)
    host.resize_screen(w, h)
    mats = []
    for position, orientation in synthetic.orbit_poses(n, centre=(0.0, 0.0, -6.0)):
        host.set_camera_pose(position, orientation)
        mats.append(np.asarray(host.camera()[0], dtype=np.float32).reshape(16))
    dev = native.DeviceScene(host.desc(), 0)
    single = native.DeviceScene(host.desc(), 0)
    try:
        def render(d, buf, **kw):
            for attempt in range(5):
                d.render_device(buf.data_ptr(), **kw)
                try:
                    return d.last_ray_counts()
                except native.RtError:
                    if attempt == 4:
                        raise
        for shard_count in (1, 3):
            kw = {} if shard_count == 1 else {"shard_index": 1, "shard_count": 3}
            dev.set_views(np.stack(mats), w, h)
            per_view = dev.shard_bytes(shard_count=shard_count) if shard_count > 1 else h * w * 3
            together = torch.zeros(n * per_view, dtype=torch.uint8, device="cuda")
            alone = torch.zeros(per_view, dtype=torch.uint8, device="cuda")
            total = render(dev, together, **kw)
            got = together.cpu().numpy().reshape(n, per_view)
            rays = np.zeros(5, dtype=np.int64)
            for v in range(n):
                single.set_camera(mats[v], w, h)
                rays += render(single, alone, **kw)[:5].astype(np.int64)
                assert np.array_equal(got[v], alone.cpu().numpy()), (shard_count, v)
            assert total[:5].astype(np.int64).tolist() == rays.tolist()
    finally:
        dev.close()
        single.close()
        host.close()


def test_five_4k_views_in_one_pass(scenes_dir, model_dir):
    """Five 3840x2160 views are 41 M pixels in one pass - more than the 2^25 a ray's pixel field once held (now 2^28).  Every
    view's bytes equal the frame rendered on its own."""
    import torch
    from rust_raytracer_amd import synthetic
    w, h = 3840, 2160
    host = native.HostScene.load(os.path.join(scenes_dir, "reflective_bunnies.json"), model_dir)
    host.resize_screen(w, h)
    mats = []
    for position, orientation in synthetic.orbit_poses(5, centre=(0.0, 0.0, -6.0)):
        host.set_camera_pose(position, orientation)
        mats.append(np.asarray(host.camera()[0], dtype=np.float32).reshape(16))
    dev = native.DeviceScene(host.desc(), 0)
    single = native.DeviceScene(host.desc(), 0)
    try:
        together = torch.zeros(5 * h * w * 3, dtype=torch.uint8, device="cuda")
        alone = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda")
        dev.set_views(np.stack(mats), w, h)
        for attempt in range(4):
            dev.render_device(together.data_ptr())
            try:
                total = dev.last_ray_counts()
                break
            except native.RtError:
                if attempt == 3:
                    raise
        got = together.cpu().numpy().reshape(5, h, w, 3)
        rays = 0
        for v in range(5):
            single.set_camera(mats[v], w, h)
            for attempt in range(4):
                single.render_device(alone.data_ptr())
                try:
                    rays += int(single.last_ray_counts()[0])
                    break
                except native.RtError:
                    if attempt == 3:
                        raise
            assert np.array_equal(got[v], alone.cpu().numpy().reshape(h, w, 3)), v
        assert int(total[0]) == rays
    finally:
        dev.close()
        single.close()
        host.close()


def test_shadow_rays_nobody_needs_are_not_traced(scenes_dir, model_dir, monkeypatch):
    """A shadow ray whose verdict cannot change the Phong sum - the surface faces away from the light and the mirrored light
    direction away from the viewer, or the colour products are zero - is counted but not traced (RT_CTR_SHADOW_NOT_TRACED).
    The picture must be the same bytes as with every shadow ray traced (RT_TRACE_MOOT_SHADOW_RAYS=1 at scene creation),
    the ray counts the reference's; and when the caller asks for the per-pixel flags, which report the verdicts of the
    primary hits themselves, those rays are traced and the flags are the oracle's."""
    import torch
    s = load(scenes_dir, model_dir, "reflective_bunnies", 960, 540)
    m, w, h = s.camera()
    want = s.render()
    monkeypatch.setenv("RT_TRACE_MOOT_SHADOW_RAYS", "1")
    every = native.DeviceScene(s.desc(), 0)
    monkeypatch.delenv("RT_TRACE_MOOT_SHADOW_RAYS")
    dev = native.DeviceScene(s.desc(), 0)
    try:
        out = [torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda") for _ in range(2)]
        counts = []
        for k, d in enumerate((every, dev)):
            d.set_camera(m, w, h)
            d.render_device(out[k].data_ptr())
            counts.append(d.last_ray_counts())
        assert torch.equal(out[0], out[1])
        assert counts[0][:5].tolist() == counts[1][:5].tolist() == [int(x) for x in want["counters"][:5]]
        assert counts[0][abi.RT_CTR_SHADOW_NOT_TRACED] == 0
        assert counts[1][abi.RT_CTR_SHADOW_NOT_TRACED] > counts[1][abi.RT_CTR_SHADOW] // 10
        assert counts[1][abi.RT_CTR_WALK_NODE_VISITS] < counts[0][abi.RT_CTR_WALK_NODE_VISITS]
        got = dev.render(aux=True)                                        # with the flags: the primary hits' shadow rays are traced
        assert_parity(got, want)
        skipped_with_flags = int(got["counters"][abi.RT_CTR_SHADOW_NOT_TRACED])
        assert 0 < skipped_with_flags < counts[1][abi.RT_CTR_SHADOW_NOT_TRACED]
    finally:
        every.close()
        dev.close()


def test_stray_rays_switch_the_exact_path_on(scenes_dir, model_dir):
    """The kernel that traces rays with non-finite components in the reference's own order is left out of the launch
    sequence until a frame turns out to hold such rays (here: a camera matrix with a NaN in it, so that every primary
    ray has a NaN component).  That frame is reported as incomplete once, like a list overflow; from the next frame on
    the kernel runs and the picture is the one rt_render (which redoes such a frame itself) delivers."""
    import torch
    s = load(scenes_dir, model_dir, "reflective_bunnies", 320, 180)
    m, w, h = s.camera()
    m = np.array(m, dtype=np.float32)
    m[1] = np.nan
    s.set_camera_matrix(m, w, h)
    dev = native.DeviceScene(s.desc(), 0)
    fresh = native.DeviceScene(s.desc(), 0)
    try:
        dev.set_camera(m, w, h)
        fresh.set_camera(m, w, h)
        side = torch.cuda.Stream()
        torch.cuda.set_stream(side)
        out = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda")
        dev.render_device(out.data_ptr(), max_depth=16, stream=side.cuda_stream)
        with pytest.raises(native.RtError) as e:
            dev.last_ray_counts()
        assert e.value.code == abi.RT_ERR_OUT_OF_RANGE and "exact path" in str(e.value)
        dev.render_device(out.data_ptr(), max_depth=16, stream=side.cuda_stream)
        counts = dev.last_ray_counts()
        got = fresh.render(max_depth=16, aux=True)                       # redone inside rt_render
        want = s.render(max_depth=16)
        assert_parity(got, want)
        assert np.array_equal(out.cpu().numpy().reshape(h, w, 3), got["rgb"])
        assert counts[:5].tolist() == want["counters"][:5].tolist()
    finally:
        dev.close()
        fresh.close()


def test_no_lights_and_empty_scene(scenes_dir, model_dir):
    scene = json.load(open(os.path.join(scenes_dir, "cornell_box.json")))
    scene["lights"] = []
    s = orc.OracleScene.from_text(json.dumps(scene))
    s.resize(160, 90)
    got, want = render_both(s)                       # Phong shade returns None -> background (raytracer.rs:76-83)
    assert_parity(got, want)
    for key in ("planes", "spheres", "triangles"):
        scene[key] = []
    s = orc.OracleScene.from_text(json.dumps(scene))
    s.resize(160, 90)
    got, want = render_both(s)
    assert_parity(got, want)
    assert not got["flags"].any()


def test_rgba_and_repeatability(scenes_dir, model_dir):
    s = load(scenes_dir, model_dir, "reflective_bunnies", 256, 144)
    m, w, h = s.camera()
    dev = native.DeviceScene(s.desc(), 0)
    try:
        dev.set_camera(m, w, h)
        a = dev.render(channels=4)["rgb"]
        b = dev.render(channels=4)["rgb"]
        c = dev.render(channels=3)["rgb"]
    finally:
        dev.close()
    assert np.array_equal(a, b)                      # integer accumulation: no run-to-run variation
    assert np.array_equal(a[..., :3], c)
    assert (a[..., 3] == 255).all()                  # web_app/wasm/src/color.rs:10-21


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_tile_shards_assemble_to_the_same_frame(scenes_dir, model_dir, shards):
    """rt_render_device with shard_index / shard_count + rt_assemble_shards: byte-identical to the
    unsharded frame for any number of shards (here all shards are rendered by the one GPU)."""
    import torch
    from rust_raytracer_amd import sharding
    s = load(scenes_dir, model_dir, "reflective_bunnies", 333, 245)
    m, w, h = s.camera()
    dev = native.DeviceScene(s.desc(), 0)
    try:
        dev.set_camera(m, w, h)
        whole = dev.render(channels=3)["rgb"]
        side = torch.cuda.Stream()                 # a real stream handle: NULL would select the handle's own stream
        torch.cuda.set_stream(side)
        stream = side.cuda_stream
        nbytes = dev.shard_bytes(shards)
        assert nbytes == sharding.shard_bytes(w, h, shards)
        gathered = torch.zeros(shards * nbytes, dtype=torch.uint8, device="cuda")
        for r in range(shards):
            part = gathered[r * nbytes:(r + 1) * nbytes]
            dev.render_device(part.data_ptr(), shard_index=r, shard_count=shards, stream=stream)
            dev.last_ray_counts()                  # raises if a ray list overflowed (the frame would be incomplete)
            assert np.array_equal(part.cpu().numpy(), sharding.extract_shard(whole, r, shards))
        frame = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda")
        dev.assemble_shards(gathered.data_ptr(), frame.data_ptr(), shards, stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(frame.cpu().numpy().reshape(h, w, 3), whole)
        full = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda")
        dev.render_device(full.data_ptr(), stream=stream)                     # shard_count 1: row-major frame
        torch.cuda.synchronize()
        assert np.array_equal(full.cpu().numpy().reshape(h, w, 3), whole)
    finally:
        dev.close()


def test_views_rendered_together(scenes_dir, model_dir):
    """rt_scene_set_views: five camera poses rendered by one call give, view by view, exactly what five calls with
    rt_scene_set_camera give (pixels, ids, flags; ray counts add up), as shards too."""
    import torch
    from rust_raytracer_amd import synthetic
    host = native.HostScene.load(os.path.join(scenes_dir, "reflective_bunnies.json"), model_dir)
    host.resize_screen(333, 117)
    dev = native.DeviceScene(host.desc(), 0)
    try:
        mats, single = [], []
        for position, orientation in synthetic.orbit_poses(5, centre=(0.0, 0.0, -6.0)):
            host.set_camera_pose(position, orientation)
            m, w, h = host.camera()
            mats.append(np.asarray(m, dtype=np.float32).reshape(16))
            dev.set_camera(m, w, h)
            single.append(dev.render(aux=True))
        dev.set_views(np.stack(mats), w, h)
        together = dev.render(aux=True)
        assert together["rgb"].shape == (5, h, w, 3)
        for v in range(5):
            for key in ("rgb", "prim_id", "instance_id", "flags"):
                assert np.array_equal(together[key][v], single[v][key]), (v, key)
        assert together["counters"][:5].tolist() == sum(np.asarray(x["counters"][:5], dtype=np.int64) for x in single).tolist()
        assert len({x["rgb"].tobytes() for x in single}) == 5                  # the poses do differ
        # the same views as three shards: every rank's buffer holds its five shards, view after view
        shards = 3
        side = torch.cuda.Stream()
        torch.cuda.set_stream(side)
        nbytes = dev.shard_bytes(shards)
        gathered = torch.zeros(shards * 5 * nbytes, dtype=torch.uint8, device="cuda")
        for r in range(shards):
            dev.render_device(gathered[r * 5 * nbytes:].data_ptr(), shard_index=r, shard_count=shards, stream=side.cuda_stream)
            dev.last_ray_counts()
        frames = torch.zeros(5 * h * w * 3, dtype=torch.uint8, device="cuda")
        dev.assemble_shards(gathered.data_ptr(), frames.data_ptr(), shards, stream=side.cuda_stream)
        full = torch.zeros(5 * h * w * 3, dtype=torch.uint8, device="cuda")
        dev.render_device(full.data_ptr(), stream=side.cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(frames.cpu().numpy().reshape(5, h, w, 3), together["rgb"])
        assert np.array_equal(full.cpu().numpy().reshape(5, h, w, 3), together["rgb"])
        # fewer views on the same handle (its buffers, made for five, stay), then one, then five again
        for n in (2, 1, 5):
            dev.set_views(np.stack(mats[5 - n:]), w, h)
            part = dev.render(aux=True)
            for key in ("rgb", "prim_id", "instance_id", "flags"):
                got = part[key].reshape((n,) + single[0][key].shape)
                assert all(np.array_equal(got[v], single[5 - n + v][key]) for v in range(n)), (n, key)
        with pytest.raises(native.RtError):
            dev.set_views(np.zeros((abi.RT_MAX_VIEWS + 1, 16), dtype=np.float32), w, h)
    finally:
        dev.close()
        host.close()


def test_kernel_timing_mode(scenes_dir, model_dir):
    """rt_set_kernel_timing / rt_last_kernel_ms: per-kernel HIP-event times of one frame, launch counts per depth,
    and no effect on the image."""
    s = load(scenes_dir, model_dir, "reflective_bunnies", 320, 180)
    m, w, h = s.camera()
    dev = native.DeviceScene(s.desc(), 0)
    try:
        dev.set_camera(m, w, h)
        plain = dev.render(max_depth=4)["rgb"]
        with pytest.raises(native.RtError):
            dev.last_kernel_ms()                    # the last frame carried no marks
        dev.set_kernel_timing(True)
        timed = dev.render(max_depth=4)["rgb"]
        k = dev.last_kernel_ms()
        trace = dev.last_trace_ms()
        dev.set_kernel_timing(False)
    finally:
        dev.close()
    assert np.array_equal(plain, timed)
    # depths 0..4 (the last one only holds the shadow rays of the deepest hits); after depth 0 rt_shade sets up the rays
    # it makes itself (scenes with few instances), so rt_setup runs once
    assert {n: c for n, (ms, c) in k.items()} == {"rt_setup": 1, "rt_traverse": 5, "rt_shade": 5}
    assert all(ms > 0.0 for ms, c in k.values())
    assert sum(ms for ms, c in k.values()) <= trace * 1.05


def test_hip_renderer_mirrors_gpu_renderer(scenes_dir, model_dir):
    """HipRenderer has GpuRenderer's four methods (gpu_renderer.rs:44-68, 280, 356-364): new / render into an
    RGBA canvas / resize_screen / turn_camera; each frame equals the oracle's after the same calls."""
    from rust_raytracer_amd import HipRenderer
    path = os.path.join(scenes_dir, "reflective_bunnies.json")
    r = HipRenderer(path, model_dir, width=256, height=144)
    s = orc.OracleScene.from_file(path, model_dir)
    try:
        s.resize(256, 144)
        canvas = np.zeros((144, 256, 4), dtype=np.uint8)
        assert r.render(canvas) is canvas
        want = s.render(channels=4, aux=False)["rgb"]
        assert (np.abs(canvas.astype(int) - want.astype(int)).max(axis=2) <= 1).mean() >= 0.999
        assert (canvas[..., 3] == 255).all()
        r.turn_camera(10.0, 20.0, 60.0, 5.0)
        s.turn_camera(10.0, 20.0, 60.0, 5.0)
        r.resize_screen(200, 120)
        s.resize(200, 120)
        assert r.size == (200, 120)
        got = r.render(channels=3)
        want = s.render(aux=False)["rgb"]
        assert got.shape == (120, 200, 3)
        assert (np.abs(got.astype(int) - want.astype(int)).max(axis=2) <= 1).mean() >= 0.999
    finally:
        r.close()


@pytest.mark.parametrize("block", range(4))
def test_random_scenes(tmp_path_factory, block):
    """Fuzz: random small scenes - every primitive kind, point and directional lights or none, the three material
    types, instances with arbitrary rotations and very non-uniform scales, both winding orders, both backgrounds.
    Ids, flags and ray counts must agree exactly with the oracle, colours within the usual tolerance."""
    from rust_raytracer_amd import synthetic
    d = str(tmp_path_factory.mktemp("fuzz_models"))
    worst = 1.0
    for seed in range(block * 12, block * 12 + 12):
        s = orc.OracleScene.from_text(synthetic.random_scene(d, seed), d)
        got, want = render_both(s, max_depth=6)
        try:
            # the generator keeps refraction off the meshes and free triangles, so the colour bar is the full one
            worst = min(worst, assert_parity(got, want, min_within_1=0.999)[1])
        except AssertionError as e:
            raise AssertionError("seed %d: %s" % (seed, e))
    print("random scenes %d..%d: worst share of pixels within 1 LSB %.5f" % (block * 12, block * 12 + 11, worst))


def test_shadow_ray_elision_with_odd_materials(tmp_path_factory):
    """Whether a shadow ray's verdict can matter is decided from the colour products and the shininess (shadow_is_moot in
    rt_kernels.hip): zero products make every verdict moot, infinite or NaN ones none (0 * inf is NaN, and radiance(),
    raytracer.rs:137-153, would compute just that), a shininess of zero or below turns 0^s into 1 or infinity.  The fuzz
    scenes with such materials and lights written into the scene description in place: flags, ids and counts are the
    oracle's, the picture within the usual bar, and the same bytes with and without the per-pixel flags (render_both)."""
    from rust_raytracer_amd import synthetic
    d = str(tmp_path_factory.mktemp("odd_materials"))
    inf, nan = float("inf"), float("nan")
    spoil = [lambda m: m.diffuse.__setitem__(0, inf),
             lambda m: (m.specular.__setitem__(1, inf), setattr(m, "shininess", 0.0)),
             lambda m: setattr(m, "shininess", -2.0),
             lambda m: [m.diffuse.__setitem__(c, 0.0) or m.specular.__setitem__(c, 0.0) for c in range(3)],
             lambda m: m.specular.__setitem__(2, nan),
             lambda m: setattr(m, "shininess", inf),
             lambda m: [m.diffuse.__setitem__(c, 0.0) for c in range(3)],
             lambda m: (m.specular.__setitem__(0, -0.0), m.specular.__setitem__(1, 0.0), m.specular.__setitem__(2, 0.0), setattr(m, "shininess", 0.0))]
    skipped = 0
    for seed in range(500, 512):
        s = orc.OracleScene.from_text(synthetic.random_scene(d, seed), d)
        desc = s.desc()
        for k in range(desc.n_materials):
            if (seed + k) % 3 != 2:
                spoil[(seed + k) % len(spoil)](desc.materials[k])                 # the oracle's own arrays, in place
        for k in range(desc.n_lights):
            if (seed + k) % 4 == 0:
                for c in range(3):
                    desc.lights[k].diffuse[c] = 0.0
            if (seed + k) % 5 == 0:
                desc.lights[k].specular[0] = inf
        got, want = render_both(s, max_depth=5)
        try:
            assert_parity(got, want, min_within_1=0.999)
        except AssertionError as e:
            raise AssertionError("seed %d: %s" % (seed, e))
        skipped += int(got["counters"][abi.RT_CTR_SHADOW_NOT_TRACED])
    assert skipped > 0


@pytest.mark.parametrize("block", range(3))
def test_spoiled_random_scenes(tmp_path_factory, block):
    """The fuzz scenes spoiled on purpose (synthetic.spoiled_scene): duplicated instances, spheres and planes (ties in t
    go to the earlier one), singular instance matrices (a NaN candidate is only taken when nothing came before it, and
    then blocks what comes after), zero radius, zero normals, collapsed triangles, a light at the camera."""
    from rust_raytracer_amd import synthetic
    d = str(tmp_path_factory.mktemp("spoiled_models"))
    seen = set()
    worst = 1.0
    for seed in range(block * 40, block * 40 + 40):
        text, what = synthetic.spoiled_scene(d, seed)
        seen.update(what)
        got, want = render_both(orc.OracleScene.from_text(text, d), max_depth=5)
        try:
            worst = min(worst, assert_parity(got, want, min_within_1=0.999)[1])
        except AssertionError as e:
            raise AssertionError("seed %d %s: %s" % (seed, what, e))
    print("spoiled scenes %d..%d: worst share of pixels within 1 LSB %.5f" % (block * 40, block * 40 + 39, worst))
    assert {"duplicate instance", "zero scale", "duplicate sphere", "duplicate plane"} <= seen


def test_mixed_scenes_with_result_only_entries(tmp_path_factory):
    """synthetic.mixed_scene: spoiled scenes with extreme cameras, lights and instance transforms and a triangle soup as
    a second mesh.  Seed 9019 is the frame that once faulted the GPU: rays traced on the exact path leave list entries
    that only record a result, and the BVH walk took them for walk items (with a null ray, through the scene's first
    mesh, on a stack sized for the shallower meshes the scene instances).  Seed 30922 holds a mesh of one triangle - its
    root is a leaf - next to an instance with a singular matrix, seen by an orthographic camera 5e7 units away: the rays
    go down the exact path, where the mesh's only box, the root's, once went untested (the reference tests every node's
    own box, intersect.rs:247-250), and so far out the box test fails where the triangle test passes.
    tests/test_kernels_on_cpu.py renders the same seeds under AddressSanitizer."""
    from rust_raytracer_amd import synthetic
    d = str(tmp_path_factory.mktemp("mixed_models"))
    rendered = 0
    for seed in [9019, 9020, 9023, 30922] + list(range(70000, 70030)):
        text, depth, what = synthetic.mixed_scene(d, seed)
        w, h = json.loads(text)["camera"]["screen_dimensions"]
        try:
            s = orc.OracleScene.from_text(text, d)
        except Exception:
            continue                                  # a soup the BVH builder refuses (the reference panics on it as well)
        got, want = render_both(s, max_depth=depth)
        try:
            assert_parity(got, want, min_within_1=0.98 if w * h > 2000 else 0.9)
        except AssertionError as e:
            raise AssertionError("seed %d %s: %s" % (seed, what, e))
        rendered += 1
    assert rendered >= 25


def test_tree_whose_boxes_do_not_nest(scenes_dir, model_dir, tmp_path_factory):
    """synthetic.shrink_bvh_boxes: every third node's box loses half of its extent, so child boxes stick out of their
    parents' and triangles out of their leaves'.  The reference tests every box as it stands (intersect.rs:245-250); the
    library must not fold such a level away, prune by distance or cull by the root box on such a mesh - on the headline
    scene and on an instance grid that walks the range tree."""
    from rust_raytracer_amd import synthetic
    d = str(tmp_path_factory.mktemp("shrunk_models"))
    for s in (load(scenes_dir, model_dir, "reflective_bunnies", 320, 180),
              orc.OracleScene.from_text(synthetic.instance_grid_scene(d, 5, 3, 2, 240, 135), d)):
        intact = s.render(max_depth=5)["prim_id"]
        assert synthetic.shrink_bvh_boxes(s.desc()) > 0                 # in place: the oracle and the device see the same spoiled tree
        got, want = render_both(s, max_depth=5)
        assert_parity(got, want)
        assert (want["prim_id"] != intact).sum() > 100                  # the spoiled boxes do change what is hit


def test_degenerate_instance_takes_the_exact_path(tmp_path_factory):
    """An instance scaled to zero along one axis: its inverse matrices are what glm::inverse makes of a singular
    matrix, no culling box can be trusted, and the local rays have NaN directions.  Every ray then goes through
    the kernel that follows the reference's own evaluation order - and must still agree with the oracle."""
    from rust_raytracer_amd import synthetic
    d = str(tmp_path_factory.mktemp("degenerate_models"))
    scene = json.loads(synthetic.instance_grid_scene(d, 3, 2, 2, 160, 90))
    scene["mesh_instances"][4]["scale"] = {"x": 0.0, "y": 1.0, "z": 1.0}
    s = orc.OracleScene.from_text(json.dumps(scene), d)
    got, want = render_both(s, max_depth=5)
    assert_parity(got, want, min_within_1=0.995)


def test_native_app_writes_the_same_picture(scenes_dir, model_dir, tmp_path):
    """tools/native_app.cpp - main.rs with paint_scene replaced by the C++ rt::HipRenderer of include/rt_renderer.hpp -
    renders a scene file at its own screen_dimensions and writes the PNG the Python binding's render would."""
    import subprocess
    from PIL import Image
    app = os.path.join(os.path.dirname(scenes_dir), "rust_raytracer_amd", "native_app")
    scene = json.load(open(os.path.join(scenes_dir, "reflective_bunnies.json")))
    scene["camera"]["screen_dimensions"] = [384, 216]
    path = str(tmp_path / "small.json")
    json.dump(scene, open(path, "w"))
    out = str(tmp_path / "render.png")
    r = subprocess.run([app, path, model_dir, out, "7", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Parsing took" in r.stdout and "Rendering took" in r.stdout
    host = native.HostScene.load(path, model_dir)
    m, w, h = host.camera()
    dev = native.DeviceScene(host.desc(), 0)
    try:
        dev.set_camera(m, w, h)
        want = dev.render(max_depth=7)
        assert (w, h) == (384, 216)
        assert np.array_equal(np.asarray(Image.open(out)), want["rgb"])
        assert ("%d rays" % int(want["counters"][0])) in r.stdout
        # the three further frames (rt::HipRenderer::render_drags_rgb: one call, three views) against one render each
        for k in range(3):
            host.turn_camera(0.0, 0.0, 40.0, 0.0)
            m, w, h = host.camera()
            dev.set_camera(m, w, h)
            assert np.array_equal(np.asarray(Image.open("%s.%d.png" % (out, k))), dev.render(max_depth=7)["rgb"]), k
    finally:
        dev.close()
        host.close()


def test_multi_device_renderer(scenes_dir, model_dir, tmp_path):
    """rt_multi_*: one process, the scene replicated per device, tiles dealt round-robin, one exchange, stitched on the
    first device.  A one-GPU box can run the RCCL path with a single device (ncclCommInitAll + ncclAllGather over one
    rank) and the several-shard schedule with every shard on device 0; both must give rt_render's bytes and ray counts."""
    import ctypes as C
    import subprocess
    from PIL import Image
    from rust_raytracer_amd import synthetic
    path = os.path.join(scenes_dir, "reflective_bunnies.json")
    host = native.HostScene.load(path, model_dir)
    host.resize_screen(333, 245)
    m, w, h = host.camera()
    dev = native.DeviceScene(host.desc(), 0)
    try:
        dev.set_camera(m, w, h)
        want = dev.render(max_depth=7)
        mats = []
        for position, orientation in synthetic.orbit_poses(3, centre=(0.0, 0.0, -6.0)):
            host.set_camera_pose(position, orientation)
            mats.append(np.asarray(host.camera()[0], dtype=np.float32).reshape(16))
        dev.set_views(np.stack(mats), w, h)
        want_views = dev.render(max_depth=7)
        for ids in ([0], [0, 0], [0, 0, 0, 0, 0]):
            multi = native.MultiDeviceScene(host.desc(), ids)
            try:
                multi.set_camera(m, w, h)
                got = multi.render(max_depth=7)
                assert np.array_equal(got["rgb"], want["rgb"]), ids
                assert got["counters"][:5].tolist() == want["counters"][:5].tolist()
                assert got["ms"] > 0.0
                rgba = multi.render(max_depth=7, channels=4)["rgb"]
                assert np.array_equal(rgba[..., :3], want["rgb"]) and (rgba[..., 3] == 255).all()
                multi.set_views(np.stack(mats), w, h)
                got = multi.render(max_depth=7)
                assert np.array_equal(got["rgb"], want_views["rgb"]), ids
                assert got["counters"][:5].tolist() == want_views["counters"][:5].tolist()
                # seven cameras in passes of three, three and one, the exchange of a pass behind the tracing of the next
                seven = np.stack([mats[k % 3] for k in range(7)])
                frames = multi.render_views(seven, 3, max_depth=7)
                assert all(np.array_equal(frames[k], want_views["rgb"][k % 3]) for k in range(7)), ids
                assert np.array_equal(multi.render(max_depth=7)["rgb"], want_views["rgb"]), ids      # back at the views set before
                with pytest.raises(native.RtError):                                                  # the renderer deals the tiles itself
                    check_params = abi.RtRenderParams(7, abi.RT_FORMAT_RGB8, 1, 2, 0, 0, None)
                    native.check(native.lib().rt_multi_render(multi._h, C.byref(check_params), frames.ctypes.data_as(C.c_void_p)))
            finally:
                multi.close()
        with pytest.raises(native.RtError):
            native.MultiDeviceScene(host.desc(), [0, 99])                 # no such device
    finally:
        dev.close()
        host.close()
    # the C++ front end: native_app --gpus 1 (RCCL, one rank) and --rehearse-gpus 3 write the picture native_app writes
    app = os.path.join(os.path.dirname(scenes_dir), "rust_raytracer_amd", "native_app")
    scene = json.load(open(path))
    scene["camera"]["screen_dimensions"] = [300, 170]
    small = str(tmp_path / "small.json")
    json.dump(scene, open(small, "w"))
    pictures = []
    for extra in ([], ["--gpus", "1"], ["--rehearse-gpus", "3"]):
        out = str(tmp_path / ("render%d.png" % len(pictures)))
        r = subprocess.run([app] + extra + [small, model_dir, out, "6", "2"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        pictures.append([np.asarray(Image.open(out))] + [np.asarray(Image.open("%s.%d.png" % (out, k))) for k in range(2)])
    for other in pictures[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(pictures[0], other))


def test_api_errors(scenes_dir, model_dir):
    """Argument validation of the C ABI: error codes with a message instead of the reference's panics."""
    import ctypes as C
    s = load(scenes_dir, model_dir, "cornell_box", 64, 48)
    L = native.lib()
    dev = native.DeviceScene(s.desc(), 0)
    try:
        out = np.zeros((48, 64, 3), dtype=np.uint8)
        p = abi.RtRenderParams(10, abi.RT_FORMAT_RGB8, 0, 1, 0, 0, None)
        assert L.rt_render(dev._h, C.byref(p), out.ctypes.data, None) == abi.RT_ERR_INVALID_ARGUMENT     # camera not set
        assert b"camera" in L.rt_last_error()
        m, w, h = s.camera()
        dev.set_camera(m, w, h)
        for bad in (abi.RtRenderParams(64, abi.RT_FORMAT_RGB8, 0, 1, 0, 0, None),        # max_depth beyond the limit
                    abi.RtRenderParams(10, 7, 0, 1, 0, 0, None),                         # unknown output format
                    abi.RtRenderParams(10, abi.RT_FORMAT_RGB8, 0, 1, 5, 0, None),        # unknown mode
                    abi.RtRenderParams(10, abi.RT_FORMAT_RGB8, 0, 1, 0, 1, None)):       # reserved field used
            assert L.rt_render(dev._h, C.byref(bad), out.ctypes.data, None) == abi.RT_ERR_INVALID_ARGUMENT
            assert L.rt_last_error() != b""
        shard = abi.RtRenderParams(10, abi.RT_FORMAT_RGB8, 3, 3, 0, 0, None)             # shard_index >= shard_count
        n = C.c_uint64()
        assert L.rt_shard_bytes(dev._h, C.byref(shard), C.byref(n)) == abi.RT_ERR_INVALID_ARGUMENT
        assert L.rt_render(dev._h, C.byref(p), None, None) == abi.RT_ERR_INVALID_ARGUMENT               # null canvas
        # views: none, too many, and so many large ones that a pixel index would not fit a ray record
        mats = np.tile(np.asarray(m, dtype=np.float32).reshape(1, 16), (abi.RT_MAX_VIEWS + 1, 1))
        fp = mats.ctypes.data_as(C.POINTER(C.c_float))
        assert L.rt_scene_set_views(dev._h, fp, 0, w, h) == abi.RT_ERR_INVALID_ARGUMENT
        assert L.rt_scene_set_views(dev._h, fp, abi.RT_MAX_VIEWS + 1, w, h) == abi.RT_ERR_INVALID_ARGUMENT
        assert L.rt_scene_set_views(dev._h, fp, 32, 4096, 4096) == 0
        assert L.rt_shard_bytes(dev._h, C.byref(p), C.byref(n)) == abi.RT_ERR_INVALID_ARGUMENT and b"pixels" in L.rt_last_error()
        dev.set_camera(m, w, h)
        assert L.rt_render(dev._h, C.byref(p), out.ctypes.data, None) == 0                                # and it still works
        assert out.any()
    finally:
        dev.close()
    # a scene whose indices point outside its arrays is refused at creation
    desc = s.desc()
    broken = abi.RtSceneDesc.from_buffer_copy(desc)
    spheres = (abi.RtSphere * desc.n_spheres).from_buffer_copy(C.string_at(desc.spheres, C.sizeof(abi.RtSphere) * desc.n_spheres))
    spheres[0].material = desc.n_materials + 3
    broken.spheres = C.cast(spheres, C.POINTER(abi.RtSphere))
    with pytest.raises(native.RtError) as e:
        native.DeviceScene(broken, 0)
    assert e.value.code == abi.RT_ERR_OUT_OF_RANGE
    with pytest.raises(native.RtError) as e:
        native.DeviceScene(desc, 99)
    assert e.value.code == abi.RT_ERR_NO_DEVICE
