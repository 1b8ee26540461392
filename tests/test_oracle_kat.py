"""The reference's own known-answer tests, run against the CPU oracle.

Cases and expected values are those of lib_raytracer/src/raytracing/intersect_tests.rs:8-161,
bvh/hull.rs:64-128 and geometry/aabb.rs:22-37; the tolerance is the reference's
(utils/mod.rs:4-7: relative error <= 2 * f32::EPSILON).
"""
import numpy as np
import pytest

from oracle import orc

EPS = float(np.finfo(np.float32).eps)


def approx_eq(a, b):
    return abs(a - b) <= 2.0 * EPS * max(abs(a), abs(b))


UNIT_SPHERE = ((0.0, 0.0, 0.0), 1.0)


@pytest.mark.parametrize("origin,direction,expected", [
    ((0, 0, -4100.0), (0, 0, 1), 4099.0),      # intersect_tests.rs:20-30
    ((0, 0, -2.0), (0, 1, 0), None),           # :33-41
    ((0, 0, -2.0), (0, 0, 1), 1.0),            # :44-54
    ((0, 0, 0.0), (0, 0, 1), 1.0),             # :57-66 (origin inside)
    ((-1, 0, -1.0), (0, 0, 1), 1.0),           # :69-78 (tangential)
    ((0, 0, -1.1), (0, 0, -1), None),          # :81-90
])
def test_ray_sphere(origin, direction, expected):
    t = orc.kat_sphere(*UNIT_SPHERE, origin, direction)
    if expected is None:
        assert t is None
    else:
        assert t is not None and approx_eq(t, expected)


TRI_V = [(-1, 1, 0), (1, 0, 0), (-1, -1, 0)]     # intersect_tests.rs:96-102
TRI_N = [(0, 0, 0)] * 3


@pytest.mark.parametrize("direction,expected", [((0, 0, 1), 2.0), ((0, 0, -1), None), ((0, 1, 0), None)])
def test_ray_triangle(direction, expected):
    t = orc.kat_triangle(TRI_V, TRI_N, (0, 0, -2.0), direction)
    if expected is None:
        assert t is None
    else:
        assert t is not None and approx_eq(t, expected)


@pytest.mark.parametrize("origin,direction,expected", [
    ((0, 0, -2.0), (0, 1, 0), None),     # parallel, intersect_tests.rs:138-143
    ((0, 0, -2.0), (0, 0, 1), 1.0),      # :146-152
    ((0, 0, 0.0), (0, 0, 1), None),      # plane behind the origin, :155-160
])
def test_ray_plane(origin, direction, expected):
    t = orc.kat_plane((0, 0, -1.0), 1.0, origin, direction)
    if expected is None:
        assert t is None
    else:
        assert t is not None and approx_eq(t, expected)


def test_hull_empty():                       # hull.rs:72-79
    assert orc.hull(np.zeros((0, 3))) is None


def test_hull_order_does_not_matter():       # hull.rs:82-105
    for pts in ([(0, 0, 0), (1, 1, 1)], [(1, 1, 1), (0, 0, 0)]):
        lo, hi = orc.hull(pts)
        assert lo.tolist() == [0, 0, 0] and hi.tolist() == [1, 1, 1]


def test_hull_of_triangle():                 # hull.rs:113-126
    lo, hi = orc.hull([(-1.0, 2.0, 3.0), (1.0, -2.0, 0.0), (0.0, 0.0, 0.0)])
    assert lo.tolist() == [-1.0, -2.0, 0.0] and hi.tolist() == [1.0, 2.0, 3.0]


def test_aabb_surface_area():                # aabb.rs:27-37
    assert orc.aabb_surface_area((0, 0, 0), (1, 1, 1)) == 6.0
    assert orc.aabb_surface_area((0, 0, 0), (2, 2, 2)) == 24.0


def test_aabb_slab_axis_parallel_ray():
    # intersect.rs:190-208 with an axis-parallel ray: 1/0 = inf, and on a face plane
    # (bound - origin) * inf = 0 * inf = NaN, which f32::min/max drop in favour of the other
    # operand (+-inf).  A ray inside the slab hits, a ray exactly in a face plane does not.
    assert orc.kat_aabb((0, 0, 0), (1, 1, 1), (0.25, 0.5, -1.0), (0, 0, 1))
    assert not orc.kat_aabb((0, 0, 0), (1, 1, 1), (0.0, 0.5, -1.0), (0, 0, 1))
    assert not orc.kat_aabb((0, 0, 0), (1, 1, 1), (1.0, 0.5, -1.0), (0, 0, 1))
    assert not orc.kat_aabb((0, 0, 0), (1, 1, 1), (2.0, 0.5, -1.0), (0, 0, 1))
    assert not orc.kat_aabb((0, 0, 0), (1, 1, 1), (0.5, 0.5, 2.0), (0, 0, 1))    # box behind the ray
    nan = float("nan")
    assert orc.kat_aabb((0, 0, 0), (1, 1, 1), (nan, nan, nan), (nan, nan, nan))   # NaN rays pass every box


def test_matrix_inverse_roundtrip():
    rng = np.random.default_rng(7)
    for _ in range(20):
        m = rng.normal(size=16).astype(np.float32)
        inv = orc.mat_inverse(m)
        prod = orc.mat_mul(m, inv).reshape(4, 4)
        assert np.allclose(prod, np.eye(4), atol=2e-3)
    assert orc.mat_inverse(np.zeros(16)).tolist() == [0.0] * 16      # glm::inverse of a singular matrix


def test_depth_map_of_a_sphere():
    """Raytracer::depth_map (raytracer.rs:21-28): brightness = 2^-t * 1.5, clamped and quantised like any colour;
    a miss shows the background.  Unit-free check on one sphere seen head-on through an orthogonal pixel ray."""
    import json
    scene = {
        "camera": {"position": {"x": 0, "y": 0, "z": 0}, "orientation_degrees": {"x": 0, "y": 0, "z": 0},
                   "screen_dimensions": [65, 65], "y_fov_degrees": 40, "z_near": 0.1, "z_far": 25,
                   "use_perspective": True, "ortho_world_height": 10},
        "background": {"SolidColor": {"r": 0, "g": 0, "b": 1}},
        "lights": [], "planes": [], "triangles": [],
        "materials": [{"name": "m", "emissive": {"r": 0, "g": 0, "b": 0}, "ambient": {"r": 0, "g": 0, "b": 0},
                       "diffuse": {"r": 0, "g": 0, "b": 0}, "specular": {"r": 0, "g": 0, "b": 0}, "shininess": 1,
                       "material_type": "Phong"}],
        "spheres": [{"center": {"x": 0, "y": 0, "z": -4}, "radius": 1, "material": "m"}],
    }
    s = orc.OracleScene.from_text(json.dumps(scene))
    out = s.render_depth_map(threads=1)["rgb"]
    # the centre pixel's ray starts on the near plane (z = -0.1) and meets the sphere at z = -3: t = 2.9
    centre = out[32, 32].astype(int)
    expect = int(min(1.0, 2.0 ** -2.9 * 1.5) * 255)
    assert abs(centre[0] - expect) <= 1 and centre[0] == centre[1] == centre[2]
    assert out[0, 0].tolist() == [0, 0, 255]                     # a corner ray misses: background
    assert out[32, 32, 0] > out[32, 20, 0] > 0                   # farther points of the sphere are darker
