"""ctypes mirror of include/rt_hip.h (struct layouts and constants only - no library is loaded here)."""
from __future__ import annotations

import ctypes as C

RT_OK = 0
RT_ERR_INVALID_ARGUMENT = -1
RT_ERR_OUT_OF_RANGE = -2
RT_ERR_BVH_TOO_DEEP = -3
RT_ERR_DEVICE = -4
RT_ERR_NO_DEVICE = -5
RT_ERR_IO = -6
RT_ERR_PARSE = -7

RT_MATERIAL_PHONG, RT_MATERIAL_REFLECT_AND_PHONG, RT_MATERIAL_REFLECT_AND_REFRACT = 0, 1, 2
RT_BACKGROUND_SOLID_COLOR, RT_BACKGROUND_COLORED_DIRECTION = 0, 1
RT_BVH_STACK_LEN = 32
RT_LEAF_TRIANGLES = 5
RT_DEFAULT_MAX_DEPTH = 10
RT_PRIM_PLANE, RT_PRIM_SPHERE, RT_PRIM_TRIANGLE, RT_PRIM_MESH_TRIANGLE = 0, 1, 2, 3
RT_PRIM_NONE = 0xFFFFFFFF
RT_INSTANCE_NONE = 0xFFFFFFFF
RT_FLAG_HIT, RT_FLAG_FRONTSIDE, RT_FLAG_REFLECTION_HIT, RT_FLAG_REFRACTION_HIT, RT_FLAG_SHADOW_LIGHT0 = 1, 2, 4, 8, 16
(RT_CTR_RAYS, RT_CTR_PRIMARY, RT_CTR_SHADOW, RT_CTR_REFLECTED, RT_CTR_TRANSMITTED, RT_CTR_AABB_TESTS,
 RT_CTR_MESH_TRIANGLE_TESTS, RT_CTR_TRIANGLE_TESTS, RT_CTR_SPHERE_TESTS, RT_CTR_PLANE_TESTS,
 RT_CTR_INSTANCE_TRANSFORMS, RT_CTR_MESH_HITS, RT_CTR_LIGHT_SHADES, RT_CTR_WALK_NODE_VISITS, RT_CTR_WALK_TRIANGLE_TESTS,
 RT_CTR_WALK_HEAVY_PAIRS, RT_CTR_SHADOW_NOT_TRACED, RT_N_COUNTERS) = range(18)
COUNTER_NAMES = ["rays", "primary", "shadow", "reflected", "transmitted", "aabb_tests", "mesh_triangle_tests",
                 "triangle_tests", "sphere_tests", "plane_tests", "instance_transforms", "mesh_hits", "light_shades",
                 "walk_node_visits", "walk_triangle_tests", "walk_heavy_pairs", "shadow_not_traced"]
(RT_LIST_USE_BYTES, RT_LIST_USE_RAY_SLOTS, RT_LIST_USE_WALK_SLOTS, RT_LIST_USE_HIT_SLOTS, RT_LIST_USE_RAYS,
 RT_LIST_USE_WALK_ITEMS, RT_LIST_USE_HITS, RT_N_LIST_USE) = range(8)
LIST_USE_NAMES = ["bytes", "ray_slots", "walk_slots", "hit_slots", "rays", "walk_items", "hits"]
RT_FORMAT_RGB8, RT_FORMAT_RGBA8 = 0, 1
RT_TILE_DIM = 32
RT_MAX_VIEWS = 256

f32, u32, i32 = C.c_float, C.c_uint32, C.c_int32


class RtLight(C.Structure):
    _fields_ = [("position", f32 * 4), ("ambient", f32 * 3), ("diffuse", f32 * 3), ("specular", f32 * 3)]


class RtMaterial(C.Structure):
    _fields_ = [("emissive", f32 * 3), ("ambient", f32 * 3), ("diffuse", f32 * 3), ("specular", f32 * 3),
                ("shininess", f32), ("material_type", u32), ("index_inner", f32), ("index_outer", f32)]


class RtPlane(C.Structure):
    _fields_ = [("normal", f32 * 3), ("distance", f32), ("material", u32)]


class RtSphere(C.Structure):
    _fields_ = [("center", f32 * 3), ("radius", f32), ("material", u32)]


class RtTriangle(C.Structure):
    _fields_ = [("vertices", (f32 * 3) * 3), ("normals", (f32 * 3) * 3), ("face_normal", f32 * 3), ("material", u32)]


class RtBvhNode(C.Structure):
    _fields_ = [("aabb_min", f32 * 3), ("aabb_max", f32 * 3), ("is_leaf", u32), ("child_left", u32),
                ("child_right", u32), ("triangle_count", u32), ("triangle_indices", u32 * RT_LEAF_TRIANGLES)]


class RtMesh(C.Structure):
    _fields_ = [("triangle_begin", u32), ("triangle_end", u32), ("node_begin", u32), ("node_end", u32),
                ("max_depth", u32)]


class RtMeshInstance(C.Structure):
    _fields_ = [("rotation_scale", f32 * 16), ("rotation_scale_inverse", f32 * 16), ("model", f32 * 16),
                ("model_inverse", f32 * 16), ("mesh", u32), ("material_override", i32)]


class RtBackground(C.Structure):
    _fields_ = [("type", u32), ("color", f32 * 3)]


class RtSceneDesc(C.Structure):
    _fields_ = [("background", RtBackground),
                ("n_lights", u32), ("n_materials", u32), ("n_planes", u32), ("n_spheres", u32),
                ("n_triangles", u32), ("n_mesh_triangles", u32), ("n_mesh_bvh_nodes", u32), ("n_meshes", u32),
                ("n_mesh_instances", u32),
                ("lights", C.POINTER(RtLight)), ("materials", C.POINTER(RtMaterial)),
                ("planes", C.POINTER(RtPlane)), ("spheres", C.POINTER(RtSphere)),
                ("triangles", C.POINTER(RtTriangle)), ("mesh_triangles", C.POINTER(RtTriangle)),
                ("mesh_bvh_nodes", C.POINTER(RtBvhNode)), ("meshes", C.POINTER(RtMesh)),
                ("mesh_instances", C.POINTER(RtMeshInstance))]


class RtAux(C.Structure):
    _fields_ = [("prim_id", C.POINTER(u32)), ("instance_id", C.POINTER(u32)), ("flags", C.POINTER(C.c_uint8)),
                ("counters", C.c_uint64 * RT_N_COUNTERS), ("hit_t", C.POINTER(f32))]


class RtRenderParams(C.Structure):
    _fields_ = [("max_depth", u32), ("output_format", u32), ("shard_index", u32), ("shard_count", u32),
                ("mode", u32), ("reserved", u32), ("stream", C.c_void_p)]


RT_MODE_RAYTRACE, RT_MODE_DEPTH_MAP = 0, 1


def desc_arrays(desc: RtSceneDesc) -> dict:
    """The arrays of a scene description as bytes, keyed by field name (for comparisons in tests)."""
    out = {}
    for name, count_name, typ in [("lights", "n_lights", RtLight), ("materials", "n_materials", RtMaterial),
                                  ("planes", "n_planes", RtPlane), ("spheres", "n_spheres", RtSphere),
                                  ("triangles", "n_triangles", RtTriangle),
                                  ("mesh_triangles", "n_mesh_triangles", RtTriangle),
                                  ("mesh_bvh_nodes", "n_mesh_bvh_nodes", RtBvhNode), ("meshes", "n_meshes", RtMesh),
                                  ("mesh_instances", "n_mesh_instances", RtMeshInstance)]:
        n = getattr(desc, count_name)
        ptr = getattr(desc, name)
        out[name] = C.string_at(ptr, n * C.sizeof(typ)) if n else b""
    out["background"] = bytes(desc.background)
    return out
