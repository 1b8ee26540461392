"""Synthetic inputs of the measurement plan that need no external assets.

``instance_grid_scene``: a G x G grid of mesh instances over M distinct meshes (icospheres whose vertices are
moved radially by xorshift32 noise, seed 0x9E3779B9 + mesh id, amplitude 1e-3), two point lights, Phong and
reflective materials alternating over the grid, a floor plane.  With M = 128 meshes of 81 920 triangles this
is the 10 M-triangle configuration; tests use a few small meshes.

``orbit_poses``: N camera poses on a circle around a point, looking at it, for the batch-of-views
configuration (radius 8, pitch -12.5 degrees, yaw step 360 / N).
"""
from __future__ import annotations

import json
import math
import os

from . import meshes

SEED0 = 0x9E3779B9


def _rgb(r, g, b):
    return {"r": r, "g": g, "b": b}


def _xyz(x, y, z):
    return {"x": x, "y": y, "z": z}


def write_grid_meshes(model_dir: str, n_meshes: int, level: int) -> list:
    """grid_mesh_<k>.obj for k < n_meshes in model_dir (existing files are kept).  Returns the names."""
    os.makedirs(model_dir, exist_ok=True)
    names = []
    for k in range(n_meshes):
        name = "grid_mesh_%d_l%d.obj" % (k, level)
        path = os.path.join(model_dir, name)
        if not os.path.exists(path):
            v, f, n = meshes.jittered_icosphere(level, (SEED0 + k) & 0xFFFFFFFF)
            tmp = path + ".tmp%d" % os.getpid()
            with open(tmp, "w") as fh:
                fh.write(meshes.obj_text(v, f, n))
            os.replace(tmp, path)
        names.append(name)
    return names


def instance_grid_scene(model_dir: str, grid: int = 4, n_meshes: int = 4, level: int = 3, width: int = 320,
                        height: int = 180, spacing: float = 2.5) -> str:
    """Scene JSON (the reference's scene-file schema) of the instance grid; the OBJ files are written to
    model_dir.  Instance (i, j) uses mesh (i * grid + j) % n_meshes, sits at (x_i, 0, z_j), is rotated by
    angles derived from (i, j) and scaled between 0.8 and 1.2; odd cells are reflective."""
    names = write_grid_meshes(model_dir, n_meshes, level)
    materials = [
        {"name": "matte", "emissive": _rgb(0.02, 0.02, 0.02), "ambient": _rgb(0.2, 0.25, 0.5), "diffuse": _rgb(0.3, 0.4, 0.7),
         "specular": _rgb(0.5, 0.5, 0.5), "shininess": 12, "material_type": "Phong"},
        {"name": "mirror", "emissive": _rgb(0, 0, 0), "ambient": _rgb(0.1, 0.1, 0.1), "diffuse": _rgb(0.3, 0.3, 0.3),
         "specular": _rgb(0.8, 0.8, 0.8), "shininess": 40, "material_type": "ReflectAndPhong"},
        {"name": "floor", "emissive": _rgb(0, 0, 0), "ambient": _rgb(0.3, 0.3, 0.25), "diffuse": _rgb(0.5, 0.5, 0.4),
         "specular": _rgb(0.2, 0.2, 0.2), "shininess": 5, "material_type": "Phong"},
    ]
    half = (grid - 1) * spacing / 2.0
    instances = []
    for i in range(grid):
        for j in range(grid):
            cell = i * grid + j
            instances.append({
                "mesh": "grid_mesh_%d" % (cell % n_meshes),
                "position": _xyz(i * spacing - half, 0.0, -6.0 - j * spacing),
                "orientation_degrees": _xyz((37 * cell) % 360, (61 * cell) % 360, (13 * cell) % 360),
                "scale": _xyz(0.8 + 0.4 * ((cell * 7) % 5) / 4.0, 0.8 + 0.4 * ((cell * 3) % 5) / 4.0, 0.8 + 0.4 * ((cell * 5) % 5) / 4.0),
                "material_override": "mirror" if (i + j) % 2 else "matte",
            })
    scene = {
        "camera": {"position": _xyz(0.0, 2.5, 1.5), "orientation_degrees": _xyz(-18.0, 0.0, 0.0),
                   "screen_dimensions": [width, height], "y_fov_degrees": 60, "z_near": 0.1, "z_far": 60,
                   "use_perspective": True, "ortho_world_height": 10},
        "background": "ColoredDirection",
        "lights": [
            {"position": {"x": 6, "y": 8, "z": 0, "w": 1},
             "color": {"ambient": _rgb(0.25, 0.25, 0.25), "diffuse": _rgb(0.6, 0.6, 0.6), "specular": _rgb(0.5, 0.5, 0.5)}},
            {"position": {"x": -7, "y": 5, "z": -14, "w": 1},
             "color": {"ambient": _rgb(0.1, 0.1, 0.1), "diffuse": _rgb(0.5, 0.5, 0.5), "specular": _rgb(0.4, 0.4, 0.4)}},
        ],
        "materials": materials,
        "planes": [{"normal": _xyz(0, 1, 0), "distance": -1.3, "material": "floor"}],
        "spheres": [],
        "triangles": [],
        "meshes": [{"name": "grid_mesh_%d" % k, "file_name": names[k], "winding_order": "CounterClockwise", "material": "matte"}
                   for k in range(n_meshes)],
        "mesh_instances": instances,
    }
    return json.dumps(scene)


def orbit_poses(n: int, centre=(0.0, 0.0, -6.0), radius: float = 8.0, pitch_degrees: float = -12.5):
    """[(position xyz, orientation radians xyz)] of n cameras on a horizontal circle of `radius` around `centre`
    (raised so that the pitch points at the centre), yaw step 360 / n.  The reference's camera looks down -z
    rotated by Ry(yaw) Rx(pitch) (transform.rs:14-21)."""
    poses = []
    pitch = math.radians(pitch_degrees)
    lift = radius * math.tan(-pitch)
    for k in range(n):
        yaw = 2.0 * math.pi * k / n
        # forward = Ry(yaw) (0, 0, -1) = (-sin yaw, 0, -cos yaw); the camera sits behind the centre
        position = (centre[0] + radius * math.sin(yaw), centre[1] + lift, centre[2] + radius * math.cos(yaw))
        poses.append((position, (pitch, yaw, 0.0)))
    return poses


def random_scene(model_dir: str, seed: int, width: int = 96, height: int = 64) -> str:
    """A random small scene in the reference's scene-file schema, for fuzzing the device path against the
    oracle: 0-3 lights (point and directional), 4-6 materials of all three types, 0-3 planes, 0-4 spheres, 0-3
    free triangles, 1-2 meshes (small icospheres, written to model_dir) and 0-6 instances with arbitrary rotations,
    non-uniform scales (down to nearly flat) and optional material overrides, either background."""
    import random
    rng = random.Random(seed)
    u = rng.uniform
    names = write_grid_meshes(model_dir, 2, 2)

    def col(lo=0.0, hi=1.0):
        return _rgb(u(lo, hi), u(lo, hi), u(lo, hi))

    kinds = ["Phong", "ReflectAndPhong", "ReflectAndRefract"]
    materials = []
    for k in range(rng.randint(4, 6)):
        mat = {"name": "m%d" % k, "emissive": col(0, 0.1), "ambient": col(0, 0.5), "diffuse": col(0, 0.8), "specular": col(0, 0.8),
               "shininess": rng.choice([0.5, 1, 8, 30]), "material_type": kinds[k % 3] if k < 3 else rng.choice(kinds)}
        if mat["material_type"] == "ReflectAndRefract":
            mat["material_type"] = {"ReflectAndRefract": {"index_inner": u(1.1, 1.7), "index_outer": 1.0}}
        materials.append(mat)
    mname = lambda: "m%d" % rng.randrange(len(materials))
    # Refraction only where the shading normal is the geometric one (spheres, planes): with the random vertex normals
    # of the triangles and the non-uniformly scaled mesh normals (transformed by rotation_scale, not its inverse
    # transpose, intersect.rs:313), Fresnel factors reach 1e20 and the reference's own float colour is noise.
    opaque = [k for k, mat in enumerate(materials) if isinstance(mat["material_type"], str)]
    oname = lambda: "m%d" % rng.choice(opaque)
    lights = []
    for _ in range(rng.randint(0, 3)):
        directional = rng.random() < 0.3
        lights.append({"position": {"x": u(-8, 8), "y": u(1, 9), "z": u(-12, 4), "w": 0 if directional else 1},
                       "color": {"ambient": col(0, 0.3), "diffuse": col(0.2, 0.8), "specular": col(0, 0.6)}})
    planes = [{"normal": _xyz(u(-0.3, 0.3), 1, u(-0.3, 0.3)), "distance": u(-3, -1), "material": mname()}]
    for _ in range(rng.randint(0, 2)):
        planes.append({"normal": _xyz(u(-1, 1), u(-1, 1), u(0.2, 1)), "distance": u(-14, -8), "material": mname()})
    if rng.random() < 0.2:
        planes = []
    spheres = [{"center": _xyz(u(-4, 4), u(-1, 3), u(-10, -3)), "radius": u(0.3, 1.5), "material": mname()}
               for _ in range(rng.randint(0, 4))]
    triangles = []
    for _ in range(rng.randint(0, 3)):
        c = (u(-4, 4), u(-1, 3), u(-10, -4))
        vs = [_xyz(c[0] + u(-2, 2), c[1] + u(-2, 2), c[2] + u(-2, 2)) for _ in range(3)]
        n = _xyz(u(-1, 1), u(-1, 1), u(0.1, 1))
        triangles.append({"vertices": vs, "normals": [n, n, n], "material": oname()})
    instances = []
    for _ in range(rng.randint(0, 6)):
        inst = {"mesh": "grid_mesh_%d" % rng.randrange(2), "position": _xyz(u(-4, 4), u(-1, 3), u(-11, -3)),
                "orientation_degrees": _xyz(u(0, 360), u(0, 360), u(0, 360)),
                "scale": _xyz(u(0.05, 1.8), u(0.3, 1.8), u(0.3, 1.8))}
        if rng.random() < 0.6:
            inst["material_override"] = oname()
        instances.append(inst)
    scene = {
        "camera": {"position": _xyz(u(-1, 1), u(0, 2), u(0, 3)), "orientation_degrees": _xyz(u(-25, 5), u(-20, 20), u(-10, 10)),
                   "screen_dimensions": [width, height], "y_fov_degrees": u(40, 80), "z_near": 0.1, "z_far": 40,
                   "use_perspective": True, "ortho_world_height": 10},
        "background": "ColoredDirection" if rng.random() < 0.5 else {"SolidColor": col()},
        "lights": lights, "materials": materials, "planes": planes, "spheres": spheres, "triangles": triangles,
        "meshes": [{"name": "grid_mesh_%d" % k, "file_name": names[k], "winding_order": rng.choice(["CounterClockwise", "Clockwise"]),
                    "material": oname()} for k in range(2)],
        "mesh_instances": instances,
    }
    return json.dumps(scene)


def spoiled_scene(model_dir: str, seed: int, width: int = 96, height: int = 64):
    """random_scene(seed) spoiled on purpose, for fuzzing the corners where the reference's arithmetic runs into ties,
    zeros, infinities and NaN: exact duplicates of an instance, a sphere, a plane (equal t: the earlier one in the
    Vec wins, utils/mod.rs:17), an instance scaled to zero along an axis (singular matrices, NaN local rays), a sphere
    of radius zero, a plane with a zero normal, a triangle with two equal vertices, a light at the camera position.
    Returns (scene JSON, list of what was done)."""
    import copy
    import random
    rng = random.Random(seed ^ 0x5F3759DF)
    scene = json.loads(random_scene(model_dir, seed, width, height))
    done = []
    instances = scene["mesh_instances"]
    if instances and rng.random() < 0.5:
        dup = copy.deepcopy(rng.choice(instances))
        if rng.random() < 0.5:
            dup.pop("material_override", None)
        instances.insert(rng.randrange(len(instances) + 1), dup)
        done.append("duplicate instance")
    if instances and rng.random() < 0.3:
        rng.choice(instances)["scale"][rng.choice("xyz")] = 0.0
        done.append("zero scale")
    if scene["spheres"] and rng.random() < 0.4:
        scene["spheres"].append(copy.deepcopy(scene["spheres"][0]))
        done.append("duplicate sphere")
    if scene["spheres"] and rng.random() < 0.2:
        scene["spheres"][-1]["radius"] = 0.0
        done.append("zero radius")
    if scene["planes"] and rng.random() < 0.4:
        scene["planes"].append(copy.deepcopy(scene["planes"][0]))
        done.append("duplicate plane")
    if scene["planes"] and rng.random() < 0.15:
        scene["planes"][-1]["normal"] = _xyz(0.0, 0.0, 0.0)
        done.append("zero normal")
    if scene["triangles"] and rng.random() < 0.3:
        scene["triangles"][0]["vertices"][1] = copy.deepcopy(scene["triangles"][0]["vertices"][0])
        done.append("collapsed triangle")
    if scene["lights"] and rng.random() < 0.2:
        p = scene["camera"]["position"]
        scene["lights"][0]["position"] = {"x": p["x"], "y": p["y"], "z": p["z"], "w": 1}
        done.append("light at the camera")
    return json.dumps(scene), done


def mixed_scene(model_dir: str, seed: int):
    """spoiled_scene(seed) pushed further, everything at once: orthographic cameras and fields of view of 1 to 180 degrees,
    light positions with w = 2, -1, 1e-3, instances 1e3 .. 1e5 away or scaled by 1e-3 / 1e3, and (half of the scenes) a
    second mesh made of a triangle soup (1 to 200 triangles from 1e-3 to 1.5 units, exact duplicates, collapsed
    triangles) instanced one to four times, now and then scaled to zero along an axis.  The first mesh of such a scene
    is deeper than a soup: a ray traced on the exact path (singular matrices) in a frame that also walks soups is what
    once made the BVH walk treat a result-only list entry as a walk item.
    Returns (scene JSON, recursion depth to render with, list of what was done)."""
    import random
    rng = random.Random(seed * 86028121 + 17)
    u = rng.uniform
    w, h = rng.choice([(96, 64), (33, 31), (130, 70), (64, 8)])
    text, what = spoiled_scene(model_dir, seed, w, h)
    scene = json.loads(text)
    cam = scene["camera"]
    if rng.random() < 0.25:
        cam["use_perspective"] = False
        cam["ortho_world_height"] = rng.choice([0.5, 10, 200])
        what.append("ortho")
    if rng.random() < 0.15:
        cam["y_fov_degrees"] = rng.choice([1.0, 150.0, 178.0, 180.0])
        what.append("fov")
    for light in scene["lights"]:
        if rng.random() < 0.25:
            light["position"]["w"] = rng.choice([2.0, -1.0, 1e-3])
            what.append("light w")
    for inst in scene["mesh_instances"]:
        r = rng.random()
        if r < 0.1:
            k = rng.choice([1e3, 1e5])
            inst["position"] = {c: inst["position"][c] * k for c in "xyz"}
            what.append("far")
        elif r < 0.2:
            k = rng.choice([1e-3, 1e3])
            inst["scale"] = {c: inst["scale"][c] * k for c in "xyz"}
            what.append("scale %g" % k)
    if rng.random() < 0.5:
        n = rng.choice([1, 6, 40, 200])
        size = rng.choice([1e-3, 0.05, 0.3, 1.5])
        lines, tris = [], []
        for k in range(n):
            c = (u(-1, 1), u(-1, 1), u(-1, 1))
            r = rng.random()
            if r < 0.1 and tris:
                tri = list(tris[rng.randrange(len(tris))])
            elif r < 0.2:
                p = [c[i] + u(-size, size) for i in range(3)]
                tri = [p, p, [c[i] + u(-size, size) for i in range(3)]]
            else:
                tri = [[c[i] + u(-size, size) for i in range(3)] for _ in range(3)]
            tris.append(tri)
        for tri in tris:
            for p in tri:
                lines.append("v %.6f %.6f %.6f" % tuple(p))
        for k in range(len(tris)):
            nn = (u(-1, 1), u(-1, 1), u(0.2, 1))
            for _ in range(3):
                lines.append("vn %.6f %.6f %.6f" % nn)
        for k in range(len(tris)):
            i = 3 * k + 1
            lines.append("f %d//%d %d//%d %d//%d" % (i, i, i + 1, i + 1, i + 2, i + 2))
        with open(os.path.join(model_dir, "soup.obj"), "w") as fh:
            fh.write("\n".join(lines) + "\n")
        opaque = [m["name"] for m in scene["materials"] if isinstance(m["material_type"], str)]
        scene["meshes"].append({"name": "soup", "file_name": "soup.obj", "winding_order": rng.choice(["Clockwise", "CounterClockwise"]),
                                "material": rng.choice(opaque)})
        for _ in range(rng.randint(1, 4)):
            scene["mesh_instances"].insert(rng.randrange(len(scene["mesh_instances"]) + 1), {
                "mesh": "soup", "position": _xyz(u(-3, 3), u(-1, 3), u(-9, -3)), "orientation_degrees": _xyz(u(0, 360), u(0, 360), u(0, 360)),
                "scale": _xyz(rng.choice([0.0, u(0.2, 2)]) if rng.random() < 0.1 else u(0.2, 2), u(0.2, 2), u(0.2, 2))})
        what.append("soup %d" % n)
    depth = rng.choice([1, 3, 5, 8])
    return json.dumps(scene), depth, what


def shrink_bvh_boxes(desc, every: int = 3, which: int = 1) -> int:
    """Spoils the BVH of a scene description in place: the box of every node whose index is `which` modulo `every`
    (inner nodes and leaves alike) loses a quarter of its extent on both sides of every axis, in float32 arithmetic
    (lo + (hi - lo) * 0.25, hi - (hi - lo) * 0.25).  Child boxes then stick out of their parents' and triangles out of
    their leaves' - a tree no builder made, as a caller may still hand one over; the reference tests every box as it
    is (intersect.rs:245-250).  tests/emu/emu_render does the same under EMU_SHRINK_BVH=1.  Returns the number of
    nodes changed."""
    import ctypes as C

    import numpy as np
    n = desc.n_mesh_bvh_nodes
    if n == 0:
        return 0
    words = C.sizeof(desc.mesh_bvh_nodes._type_) // 4
    raw = np.ctypeslib.as_array(C.cast(desc.mesh_bvh_nodes, C.POINTER(C.c_float)), shape=(n, words))
    idx = np.arange(n)[np.arange(n) % every == which]
    lo, hi = raw[idx, 0:3].copy(), raw[idx, 3:6].copy()
    w = (hi - lo).astype(np.float32)
    q = (w * np.float32(0.25)).astype(np.float32)
    raw[idx, 0:3] = (lo + q).astype(np.float32)
    raw[idx, 3:6] = (hi - q).astype(np.float32)
    return len(idx)


def pruning_stress_scene(model_dir: str, seed: int = 0):
    """A scene built to sit on the edges of the BVH walk's distance pruning (the walk may only skip what cannot win):

    * a lattice of axis-aligned squares (two triangles each, edge 0.35, so |e1||e2| = 0.1225 stays just inside the bound
      of 0.125 up to which a mesh is walked with pruning): their boxes are flat, and rays parallel to an axis through
      dyadic coordinates run exactly inside triangle planes and along box faces;
    * slivers almost parallel to the view direction: Moeller-Trumbore's determinant for a ray along -z is between 1e-5
      and 2e-5, just above the reference's rejection threshold (intersect.rs:152), where the computed t is least
      accurate; several of them behind each other so that a wrongly pruned one would change the winner;
    * point lights exactly on lattice vertices, so shadow rays end at the distance of a candidate (raytracer.rs:178-186).

    Returns (scene JSON text, [(screen_to_world column-major 16 floats, width, height)] of parallel-ray cameras that
    look along -z: one over the whole lattice on dyadic coordinates, one zoomed into the slivers)."""
    import random
    rng = random.Random(seed)
    os.makedirs(model_dir, exist_ok=True)
    v, f = [], []

    def quad(p, a, b):
        k = len(v)
        v.extend([p, tuple(p[i] + a[i] for i in range(3)), tuple(p[i] + a[i] + b[i] for i in range(3)), tuple(p[i] + b[i] for i in range(3))])
        f.extend([(k, k + 1, k + 2), (k, k + 2, k + 3)])

    e = 0.35
    for i in range(5):
        for j in range(5):
            for k in range(3):
                p = (-0.875 + i * 0.375, -0.875 + j * 0.375, -4.0 - k * 0.375)
                which = rng.randrange(3)
                if which == 0: quad(p, (e, 0, 0), (0, e, 0))          # plane z = const: hit head-on
                elif which == 1: quad(p, (e, 0, 0), (0, 0, -e))       # plane y = const: rays run inside it
                else: quad(p, (0, e, 0), (0, 0, -e))                  # plane x = const
    sliver_y = 1.25
    for k in range(6):
        s = (1.0 + 0.15 * k) * 1e-4                                   # determinant 0.1225 * s for a ray along -z
        z0 = -2.0 - 0.36 * k
        kk = len(v)
        v.extend([(-0.175, sliver_y + 3e-6 * k, z0), (0.175, sliver_y + 3e-6 * k, z0), (-0.175, sliver_y + 3e-6 * k + e * s, z0 - e)])
        f.append((kk, kk + 1, kk + 2))
    name = "pruning_stress_%d.obj" % seed
    lines = ["o stress"] + ["v %.9g %.9g %.9g" % p for p in v] + ["vn 0 0 1"]
    lines += ["f %d//1 %d//1 %d//1" % (a + 1, b + 1, c + 1) for a, b, c in f]
    with open(os.path.join(model_dir, name), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    scene = {
        "camera": {"position": _xyz(0, 0, 0), "orientation_degrees": _xyz(0, 0, 0), "screen_dimensions": [64, 64], "y_fov_degrees": 60,
                   "z_near": 0.1, "z_far": 60, "use_perspective": True, "ortho_world_height": 10},
        "background": "ColoredDirection",
        "lights": [{"position": {"x": -0.875 + 0.375 * a, "y": -0.875 + 0.375 * b, "z": -4.0 - 0.375 * c, "w": 1},
                    "color": {"ambient": _rgb(0.2, 0.2, 0.2), "diffuse": _rgb(0.5, 0.5, 0.5), "specular": _rgb(0.3, 0.3, 0.3)}}
                   for a, b, c in [(1, 2, 0), (3, 1, 1)]],
        "materials": [{"name": "matte", "emissive": _rgb(0.05, 0.02, 0.02), "ambient": _rgb(0.3, 0.3, 0.4), "diffuse": _rgb(0.5, 0.6, 0.7),
                       "specular": _rgb(0.4, 0.4, 0.4), "shininess": 9, "material_type": "Phong"},
                      {"name": "mirror", "emissive": _rgb(0, 0, 0), "ambient": _rgb(0.1, 0.1, 0.1), "diffuse": _rgb(0.2, 0.2, 0.2),
                       "specular": _rgb(0.7, 0.7, 0.7), "shininess": 30, "material_type": "ReflectAndPhong"}],
        "planes": [{"normal": _xyz(0, 0, 1), "distance": -6.0, "material": "matte"}],
        "spheres": [], "triangles": [],
        "meshes": [{"name": "stress", "file_name": name, "winding_order": "CounterClockwise", "material": "matte"}],
        "mesh_instances": [{"mesh": "stress", "position": _xyz(0, 0, 0), "orientation_degrees": _xyz(0, 0, 0), "scale": _xyz(1, 1, 1)},
                           {"mesh": "stress", "position": _xyz(0.375, 0, -1.5), "orientation_degrees": _xyz(0, 0, 0), "scale": _xyz(1, 1, 1),
                            "material_override": "mirror"}],
    }

    def parallel_camera(x0, y0, step, width, height, step_y=None):
        # pixel (x, y) -> origin (x0 + x * step, y0 + y * step_y, 0), direction (0, 0, -1)  (raytracer.rs:30-51 with an affine matrix)
        m = [0.0] * 16
        m[0] = step; m[5] = step if step_y is None else step_y; m[10] = -1.0
        m[12], m[13], m[14], m[15] = x0, y0, 0.0, 1.0
        return m, width, height

    cameras = [parallel_camera(-1.0, -1.0, 1.0 / 64.0, 160, 160),                       # dyadic coordinates: rays inside planes and box faces
               parallel_camera(-0.2, sliver_y - 8.0 / 1048576.0, 1.0 / 256.0, 96, 96, 1.0 / 1048576.0),  # 2^-20 per row across the slivers
               parallel_camera(-0.99, -0.99, 0.0123, 170, 170)]                          # a generic lattice of rays
    return json.dumps(scene), cameras


def far_origin_scene(model_dir: str, level: int = 4):
    """Nine instances of a mesh of many leaves seen through parallel rays that start 2^21 units away on every axis: at that
    distance the slab test of intersect.rs:185-226 tells none of the mesh's boxes apart (its products all round to a few
    neighbouring values), so every ray is a tube a quarter of a unit thick that passes whatever leaf box comes near it, and
    the reference tests all their triangles - the pairs the device path hands to traverse_heavy_pairs instead of walking
    them.  The first frame holds more such pairs than that code takes per depth, so the rest are walked as usual, side by
    side with it; the second looks nearly along -z from 10^7 units: for the instances that are not turned two of the three
    slabs still resolve the boxes - rays that are far away but thin, and are walked - while the turned ones see the ray's
    origin far out on two of their own axes.

    Returns (scene JSON text, [(screen_to_world column-major 16 floats, width, height)] of the two cameras)."""
    from . import meshes
    os.makedirs(model_dir, exist_ok=True)
    with open(os.path.join(model_dir, "ball.obj"), "w") as fh:
        fh.write(meshes.obj_text(*meshes.jittered_icosphere(level, 0x9E3779B9)))
    instances = []
    for i in range(3):
        for j in range(3):
            inst = {"mesh": "ball", "position": _xyz(-2.5 + 2.5 * i, -2.5 + 2.5 * j, -0.4 * (i + j)),
                    "orientation_degrees": _xyz(10 * i, 20 * j, 30), "scale": _xyz(1, 1.1 - 0.1 * i, 1)}
            if (i + j) % 2:
                inst["material_override"] = "mirror"
            instances.append(inst)
    scene = {
        "camera": {"position": _xyz(0, 0, 4), "orientation_degrees": _xyz(0, 0, 0), "screen_dimensions": [48, 48], "y_fov_degrees": 40,
                   "z_near": 0.1, "z_far": 60, "use_perspective": True, "ortho_world_height": 10},
        "background": "ColoredDirection",
        "lights": [{"position": {"x": 2.0, "y": 3.0, "z": 4.0, "w": 1},
                    "color": {"ambient": _rgb(0.2, 0.2, 0.2), "diffuse": _rgb(0.6, 0.6, 0.6), "specular": _rgb(0.3, 0.3, 0.3)}}],
        "materials": [{"name": "matte", "emissive": _rgb(0.02, 0.02, 0.05), "ambient": _rgb(0.3, 0.3, 0.4), "diffuse": _rgb(0.5, 0.6, 0.7),
                       "specular": _rgb(0.4, 0.4, 0.4), "shininess": 9, "material_type": "Phong"},
                      {"name": "mirror", "emissive": _rgb(0, 0, 0), "ambient": _rgb(0.1, 0.1, 0.1), "diffuse": _rgb(0.2, 0.2, 0.2),
                       "specular": _rgb(0.7, 0.7, 0.7), "shininess": 30, "material_type": "ReflectAndPhong"}],
        "planes": [{"normal": _xyz(0, 0, 1), "distance": -3.0, "material": "matte"}],
        "spheres": [], "triangles": [],
        "meshes": [{"name": "ball", "file_name": "ball.obj", "winding_order": "CounterClockwise", "material": "matte"}],
        "mesh_instances": instances,
    }
    # pixel (x, y) -> origin (-8 + x / 4, -8 + y / 4, 0) + 2^21 * (1, 1, 1), direction along (-1, -1, -1): multiples of the
    # spacing of float32 at 2^21 (a quarter)  (raytracer.rs:30-51 with an affine matrix)
    far = 2.0 ** 21
    m1 = [0.0] * 16
    m1[0] = 0.25; m1[5] = 0.25
    m1[8], m1[9], m1[10] = -1.0, -1.0, -1.0
    m1[12], m1[13], m1[14], m1[15] = -8.0 + far, -8.0 + far, far, 1.0
    # pixel (x, y) -> origin target(x, y) - 1e7 * (2^-10, 2^-9, -1), direction along (2^-10, 2^-9, -1), target(x, y) =
    # (-4 + x / 8, -4 + y / 8, 0); powers of two, so that the difference of the two unprojected points the direction is
    # made of survives float32 at these magnitudes
    m2 = [0.0] * 16
    m2[0] = 0.125; m2[5] = 0.125
    m2[8], m2[9], m2[10] = 2.0 ** -10, 2.0 ** -9, -1.0
    m2[12], m2[13], m2[14], m2[15] = -4.0 - 1e7 * 2.0 ** -10, -4.0 - 1e7 * 2.0 ** -9, 1e7, 1.0
    return json.dumps(scene), [(m1, 64, 64), (m2, 64, 64)]
