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
