"""Deterministic stand-in meshes.

The reference's scenes name three OBJ files (res/scenes/*.json "meshes[].file_name"):
``bunny.obj`` and ``santa.obj`` are absent from the reference tree (.MISSING_LARGE_BLOBS:1-4) and
``sphere_high.obj`` cannot travel to the GPU box.  The generators below write OBJ files of the same
record subset the reference's loader accepts (object_file.rs:63-79: ``v``, ``vn``, ``f a//b``;
counter-clockwise winding, outward vertex normals) so that every scene can be assembled anywhere.
A user-supplied file of the same name in the model directory always wins.

Only +, -, *, / and sqrt on float64 are used, and coordinates are printed with six decimals, so the
files are bit-reproducible across machines.
"""
from __future__ import annotations

import os
import tempfile

import numpy as np

__all__ = ["icosphere", "bunny_standin", "santa_standin", "uv_sphere", "obj_text", "ensure_model_dir",
           "jittered_icosphere"]


def icosphere(level: int):
    """Unit icosphere: (vertices[n,3] float64, faces[m,3] int64), m = 20 * 4**level, CCW from outside."""
    t = (1.0 + np.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0],
                  [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.sqrt((v * v).sum(1))[:, None]
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
                  [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
                  [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        n = len(v)
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        e_sorted = np.sort(e, axis=1)
        key = e_sorted[:, 0] * n + e_sorted[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a, b = uniq // n, uniq % n
        mid = v[a] + v[b]
        mid /= np.sqrt((mid * mid).sum(1))[:, None]
        v = np.concatenate([v, mid])
        m = n + inv.reshape(3, -1).T          # midpoints of edges (01, 12, 20) per face
        f = np.concatenate([
            np.stack([f[:, 0], m[:, 0], m[:, 2]], 1),
            np.stack([f[:, 1], m[:, 1], m[:, 0]], 1),
            np.stack([f[:, 2], m[:, 2], m[:, 1]], 1),
            np.stack([m[:, 0], m[:, 1], m[:, 2]], 1)])
    return v, f


def _vertex_normals(v, f):
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])      # area-weighted, outward for CCW
    n = np.zeros_like(v)
    for k in range(3):
        np.add.at(n, f[:, k], fn)
    ln = np.sqrt((n * n).sum(1))
    ln[ln == 0] = 1.0
    return n / ln[:, None]


def _bump(d, centre, sharpness):
    c = np.asarray(centre, dtype=np.float64)
    c = c / np.sqrt((c * c).sum())
    q = ((d - c) ** 2).sum(1)
    return 1.0 / (1.0 + sharpness * q) ** 2


def bunny_standin(level: int = 6):
    """A star-shaped blob with a head and two ears, about 1.5 units tall, base near y = 0.

    Stands in for the Stanford bunny the reference's og_scene_rust.json instances four times
    (res/scenes/og_scene_rust.json:335-420).  20 * 4**level triangles (81 920 at level 6; the
    full-resolution bunny has about 69 000).
    """
    d, f = icosphere(level)
    r = 0.30 + 0.0 * d[:, 0]
    r = r + 0.10 * _bump(d, (0.0, -1.0, 0.2), 1.5)          # haunches
    r = r + 0.22 * _bump(d, (0.55, 0.75, 0.0), 6.0)         # head
    r = r + 0.42 * _bump(d, (0.30, 1.0, 0.22), 60.0)        # ears
    r = r + 0.42 * _bump(d, (0.30, 1.0, -0.22), 60.0)
    r = r + 0.08 * _bump(d, (-1.0, -0.2, 0.0), 25.0)        # tail
    r = r + 0.012 * (d[:, 0] * d[:, 1] * d[:, 2]) * 8.0     # gentle asymmetry
    v = d * r[:, None]
    v[:, 1] *= 1.15
    v[:, 1] += 0.42
    v = v * 1.45         # the reference picture (renders/reflective_bunnies.png) shows bunnies of about this size
    return v, f, _vertex_normals(v, f)


def santa_standin(level: int = 6):
    """Stand-in for santa.obj (mirror_sphere.json, infinity_santa.json, santa_and_balls.json): a
    tall body with a pointed hat, about two units tall, centred on the origin."""
    d, f = icosphere(level)
    r = 0.55 + 0.0 * d[:, 0]
    r = r + 0.25 * _bump(d, (0.0, -1.0, 0.0), 2.0)
    r = r + 0.55 * _bump(d, (0.0, 1.0, 0.0), 18.0)
    r = r + 0.18 * _bump(d, (1.0, 0.1, 0.0), 20.0)
    r = r + 0.18 * _bump(d, (-1.0, 0.1, 0.0), 20.0)
    r = r + 0.10 * _bump(d, (0.0, 0.2, 1.0), 12.0)
    v = d * r[:, None]
    v[:, 1] *= 1.2
    return v, f, _vertex_normals(v, f)


def uv_sphere(level: int = 5, radius: float = 0.5):
    """Stand-in for sphere_high.obj (22 200 triangles, radius 0.5): an icosphere of radius 0.5
    (20 480 triangles at level 5)."""
    d, f = icosphere(level)
    return d * radius, f, d.copy()


def jittered_icosphere(level: int, seed: int, amplitude: float = 1e-3):
    """Icosphere with vertices moved radially by xorshift32 noise; used for the synthetic
    many-mesh scene."""
    d, f = icosphere(level)
    x = np.uint32(seed if seed else 1)
    n = len(d)
    noise = np.empty(n, dtype=np.float64)
    s = int(x)
    for i in range(n):
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        noise[i] = s / 4294967296.0
    v = d * (1.0 + amplitude * (noise - 0.5))[:, None]
    return v, f, _vertex_normals(v, f)


def obj_text(v, f, n) -> str:
    """OBJ text with one normal per vertex and ``f a//a b//b c//c`` faces."""
    out = ["o standin"]
    out += ["v %.6f %.6f %.6f" % tuple(p) for p in v]
    out += ["vn %.6f %.6f %.6f" % tuple(p) for p in n]
    out += ["f %d//%d %d//%d %d//%d" % (a, a, b, b, c, c) for a, b, c in (f + 1)]
    return "\n".join(out) + "\n"


_GENERATORS = {
    "bunny.obj": lambda: bunny_standin(6),
    "santa.obj": lambda: santa_standin(6),
    "sphere_high.obj": lambda: uv_sphere(5),
    "sphere_low.obj": lambda: uv_sphere(1),
}


def ensure_model_dir(path: str | None = None, names=None, levels: dict | None = None) -> str:
    """Make sure `path` holds every OBJ in `names` (default: all stand-ins); existing files -
    user-supplied real models included - are left alone.  `levels` overrides the subdivision
    level per file name (tests use small meshes).  Returns the directory."""
    if path is None:
        path = os.path.join(tempfile.gettempdir(), "rust_raytracer_amd_models_%d" % os.getuid())
    os.makedirs(path, exist_ok=True)
    for name in (names or _GENERATORS):
        target = os.path.join(path, name)
        if os.path.exists(target):
            continue
        if levels and name in levels:
            gen = {"bunny.obj": bunny_standin, "santa.obj": santa_standin,
                   "sphere_high.obj": uv_sphere, "sphere_low.obj": uv_sphere}[name]
            mesh = gen(levels[name])
        else:
            mesh = _GENERATORS[name]()
        tmp = target + ".tmp%d" % os.getpid()
        with open(tmp, "w") as fh:
            fh.write(obj_text(*mesh))
        os.replace(tmp, target)
    return path
