// TEST INFRASTRUCTURE - CPU oracle, not part of the shipped renderer.
//
// A restatement of the reference's render path in C++ (the reference is Rust and cannot be built here).  Pinned by
// the reference's own known answers (the intersection tests of raytracing/intersect_tests.rs, the hull and area
// tests of bvh/hull.rs and geometry/aabb.rs) and by mesh-free crops of its committed render of og_scene_rust.json;
// BVH traversal and instance transforms on meshes, for which the reference holds no fixture, are parity unpinned by
// reference outputs; tests/test_oracle_bvh_property.py ties them to the pinned triangle test instead (the same triangles
// as a mesh behind a BVH and an instance transform, and as free triangles, must give the same picture).
#pragma once
#include "orc_scene.h"

namespace orc {

// raytracing/ray.rs
struct Ray { Vec3 origin, direction; };

// raytracing/hitpoint.rs:4-12, plus the identity of what was hit (the reference keeps none).
struct Hitpoint {
    float t;
    Vec3 position;
    Vec3 hit_normal;
    Vec3 position_for_refraction;
    bool on_frontside;
    uint32_t material;
    uint32_t prim_id;      // RT_PRIM_* << 30 | index
    uint32_t instance_id;  // RT_INSTANCE_NONE unless a mesh instance was hit
};

struct Counters { uint64_t c[RT_N_COUNTERS] = {}; };

// intersect.rs - each returns false for None
bool intersect_sphere(const RtSphere& s, const Ray& ray, Hitpoint* out);
bool intersect_plane(const RtPlane& p, const Ray& ray, Hitpoint* out);
bool intersect_triangle(const RtTriangle& t, const Ray& ray, Hitpoint* out);
bool intersect_aabb(const float bmin[3], const float bmax[3], const Ray& ray);
// scene.rs:120-140
bool intersect_scene(const Scene& s, const Ray& ray, Hitpoint* out, Counters* ctr);
// Raytracer::depth_map for every pixel, raytracer.rs:21-28 (misses show the background, main.rs:58-61)
void render_depth_map(const Scene& s, int threads, int channels, uint8_t* rgb, uint64_t counters[RT_N_COUNTERS]);

struct PixelResult {
    Vec3 color;           // before quantisation
    uint32_t prim_id, instance_id;
    uint8_t flags;
    float t;              // Hitpoint::t of the primary hit, NaN without one
};

// raytracer.rs:30-51
Ray generate_primary_ray(const Scene& s, float x, float y);
// native_app/src/main.rs:56-61 for one pixel; (x, y) are the reference's screen coordinates (y up)
PixelResult trace_pixel(const Scene& s, float x, float y, uint32_t max_depth, Counters* ctr);
// color.rs:42-52
void quantize(Vec3 c, uint8_t out[3]);

// native_app/src/main.rs:46-66 over rows [row_begin, row_end) of the canvas with `threads` workers
// pulling rows from a shared counter (the reference uses rayon's par_bridge over rows).  Arrays
// may be null.  rgb has `channels` (3 or 4) bytes per pixel; alpha = 255
// (web_app/wasm/src/color.rs:10-21).
void render_frame(const Scene& s, uint32_t max_depth, int threads, uint32_t row_begin, uint32_t row_end,
                  int channels, uint8_t* rgb, uint32_t* prim_id, uint32_t* instance_id, uint8_t* flags, float* hit_t,
                  uint64_t counters[RT_N_COUNTERS]);

}  // namespace orc
