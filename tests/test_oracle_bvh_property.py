"""The oracle's mesh path against its own analytic path.

The reference holds no fixture with a mesh in it, so the BVH builder, the BVH walk and the instance transform of the
oracle are not pinned by reference outputs.  They are tied here to the part that is pinned (the ray-triangle test of
intersect_tests.rs, through the free triangles of Scene::triangles): the same triangles, once as a mesh behind a BVH and
an instance transform, once as free triangles that every ray tests one by one (scene.rs:129-131), must show the same
picture.  A walk that skips a node it should visit, or a builder that loses a triangle, shows up as a differing pixel."""
import json

import numpy as np
import pytest

from oracle import orc
from rust_raytracer_amd import meshes

W, H = 160, 120


def _xyz(p):
    return {"x": float(p[0]), "y": float(p[1]), "z": float(p[2])}


def _scene(mesh_instances, meshes_, triangles):
    grey = {"r": 0.4, "g": 0.4, "b": 0.4}
    return {
        "camera": {"position": _xyz((0.2, 0.3, 3.0)), "orientation_degrees": _xyz((-4, 3, 0)), "screen_dimensions": [W, H],
                   "y_fov_degrees": 50, "z_near": 0.1, "z_far": 30, "use_perspective": True, "ortho_world_height": 10},
        "background": {"SolidColor": {"r": 0.1, "g": 0.2, "b": 0.3}},
        "lights": [{"position": {"x": 3, "y": 4, "z": 5, "w": 1}, "color": {"ambient": grey, "diffuse": grey, "specular": grey}}],
        "materials": [{"name": "m", "emissive": {"r": 0, "g": 0, "b": 0}, "ambient": {"r": 0.3, "g": 0.2, "b": 0.2},
                       "diffuse": {"r": 0.7, "g": 0.5, "b": 0.3}, "specular": {"r": 0.5, "g": 0.5, "b": 0.5}, "shininess": 20,
                       "material_type": "ReflectAndPhong"}],
        "planes": [], "spheres": [], "triangles": triangles, "meshes": meshes_, "mesh_instances": mesh_instances,
    }


def _rounded(a):
    """What the OBJ text of meshes.obj_text holds: six decimals."""
    return np.array([[float("%.6f" % x) for x in row] for row in a])


def _render(scene, model_dir):
    s = orc.OracleScene.from_text(json.dumps(scene), model_dir)
    return s.render(10)


@pytest.mark.parametrize("level,winding", [(1, "Clockwise"), (3, "CounterClockwise")])
def test_mesh_behind_bvh_equals_free_triangles(tmp_path, level, winding):
    v, f, n = meshes.jittered_icosphere(level, 12345, amplitude=0.2)
    (tmp_path / "blob.obj").write_text(meshes.obj_text(v, f, n))
    v, n = _rounded(v), _rounded(n)
    if winding == "CounterClockwise":
        n = -n                                                                    # object_file.rs:44-48
    as_mesh = _scene([{"mesh": "blob", "position": _xyz((0, 0, 0)), "orientation_degrees": _xyz((0, 0, 0)), "scale": _xyz((1, 1, 1))}],
                     [{"name": "blob", "file_name": "blob.obj", "winding_order": winding, "material": "m"}], [])
    as_triangles = _scene([], [], [{"vertices": [_xyz(v[i]) for i in tri], "normals": [_xyz(n[i]) for i in tri], "material": "m"}
                                   for tri in f])
    a, b = _render(as_mesh, str(tmp_path)), _render(as_triangles, str(tmp_path))
    assert a["counters"][:5].tolist() == b["counters"][:5].tolist()          # the same rays were spawned
    hit = (a["flags"] & 1) != 0
    assert np.array_equal(hit, (b["flags"] & 1) != 0)
    assert 0.1 < hit.mean() < 0.9
    # The identity transform leaves the ray as it is, so both paths run the same arithmetic on the same triangle;
    # only a ray through a shared edge may pick the other neighbour (the BVH meets the triangles in another order).
    same_triangle = (a["prim_id"] & 0x3FFFFFFF) == (b["prim_id"] & 0x3FFFFFFF)
    assert same_triangle[hit].mean() > 0.995
    diff = np.abs(a["rgb"].astype(int) - b["rgb"].astype(int)).max(axis=2)
    assert (diff[same_triangle] == 0).mean() > 0.999            # (a reflected ray may meet such an edge too)
    assert (diff <= 1).mean() > 0.995


def test_instance_transform_equals_transformed_triangles(tmp_path):
    """The same through a rotation, a non-uniform scale and a translation: the free triangles are the mesh's
    triangles moved to world space in float64; the colours agree up to rounding."""
    v, f, n = meshes.jittered_icosphere(2, 777, amplitude=0.2)
    (tmp_path / "blob.obj").write_text(meshes.obj_text(v, f, n))
    v, n = _rounded(v), _rounded(n)
    position, degrees, scale = (0.3, -0.2, -0.5), (20.0, 50.0, 10.0), (1.2, 0.8, 1.0)
    inst = orc.make_instance(0, position, tuple(np.radians(degrees)), scale)
    model = np.array(inst.model, dtype=np.float64).reshape(4, 4).T                    # column-major -> [row][col]
    rs = np.array(inst.rotation_scale, dtype=np.float64).reshape(4, 4).T
    vw = v @ model[:3, :3].T + model[:3, 3]
    nw = -n @ rs[:3, :3].T         # object_file.rs:44-48 (counter-clockwise: negated); intersect.rs:313: rotation_scale, not its inverse transpose
    nw /= np.linalg.norm(nw, axis=1, keepdims=True)
    as_mesh = _scene([{"mesh": "blob", "position": _xyz(position), "orientation_degrees": _xyz(degrees), "scale": _xyz(scale)}],
                     [{"name": "blob", "file_name": "blob.obj", "winding_order": "CounterClockwise", "material": "m"}], [])
    as_triangles = _scene([], [], [{"vertices": [_xyz(vw[i]) for i in tri], "normals": [_xyz(nw[i]) for i in tri], "material": "m"}
                                   for tri in f])
    a, b = _render(as_mesh, str(tmp_path)), _render(as_triangles, str(tmp_path))
    hit_a, hit_b = (a["flags"] & 1) != 0, (b["flags"] & 1) != 0
    assert (hit_a == hit_b).mean() > 0.998                                            # silhouette pixels may flip
    both = hit_a & hit_b
    assert ((a["prim_id"] & 0x3FFFFFFF) == (b["prim_id"] & 0x3FFFFFFF))[both].mean() > 0.99
    diff = np.abs(a["rgb"].astype(int) - b["rgb"].astype(int)).max(axis=2)
    assert (diff <= 2).mean() > 0.98
