// gfx950 kernels of the frame renderer.
//
// What is computed is the reference's per-pixel path, native_app/src/main.rs:46-66 ->
// raytracer.rs (generate_primary_ray, raytrace_impl, shade, radiance, trace_shadow_ray) ->
// scene.rs:120-140 -> intersect.rs.  How it is computed is different:
//
//  * The recursion of raytracer.rs:65-135 is flattened.  The colour of a pixel is linear in the
//    colours of the child rays (reflection: child*0.8 + phong, raytracer.rs:93,129; refraction:
//    refl*k + trans*(1-k), raytracer.rs:119-120, where k depends only on the ray directions), so a
//    ray carries the product of the weights on its path ("throughput") and every ray adds
//    throughput * local colour to its pixel.  Sums are kept in 2^-40 fixed point with integer
//    atomics, so the result does not depend on the order in which rays finish.
//  * All rays of one recursion depth - and the shadow rays of the hits one depth up - form a list
//    in HBM (32 B per ray).  Each depth is a sequence of dense launches over such lists:
//      A   rt_setup       per ray: analytic primitives; per instance a world-space culling box -> the
//                         (ray, instance) pairs the ray can touch; per pair, 64 pairs at a time: the mesh-local
//                         ray and the exact root-box test -> 64-byte walk item
//      A'  rt_setup_slow  the few rays rt_setup does not treat (non-finite components, degenerate instance
//                         matrices), in the reference's exact evaluation order; only launched for scenes that need it
//      B   rt_traverse    BVH walk, one walk item per lane, lanes refilled as they finish - every lane
//                         that is busy is busy with the same code
//      C   rt_shade       closest hit -> hit point, material, child rays and hit records; then the Phong sums
//                         of the previous depth's hits, whose shadow rays were traced at this depth; then, for
//                         scenes with few instances, step A of the next depth for the rays the block has just
//                         made (rt_setup itself then only runs at depth 0)
//    Splitting by kind of work is what keeps the 64 lanes of a wave busy: in a one-kernel version
//    fewer than 5 % of the lanes had a BVH node to visit in a given round.
//  * Lists are appended to without device-wide atomics (about 10 ns each when contended, all
//    waves serialised): every block owns a segment of each output list, see rt_device.h.
//
// Every arithmetic expression that feeds a hit decision keeps the operation order of the reference
// (see the citations at each function); the file must be compiled with -ffp-contract=off and
// without fast-math so that no FMA is formed and division / sqrt stay correctly rounded.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "rt_device.h"
#ifdef RT_STATS
#include <cstdlib>
#endif

#define RT_BLOCK_THREADS 256
// Block sizes of the list kernels.  Their grids are fixed (one list segment per block); 512-thread blocks were
// measured to change nothing on the 1080p bunnies frame, so all stay at 256.
#ifndef RT_SETUP_THREADS
#define RT_SETUP_THREADS 256
#endif
#ifndef RT_SHADE_THREADS
#define RT_SHADE_THREADS 256
#endif
#define RT_MAX_LIST_THREADS 512
#define RT_WAVE 64

namespace {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 cmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
// nalgebra's 3-vector dot: (a0*b0 + a1*b1) + a2*b2
__device__ __forceinline__ float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// glm::normalize: true division by the norm, no zero guard
__device__ __forceinline__ V3 normalize3(V3 a) {
    float n = sqrtf(dot3(a, a));
    return mk(a.x / n, a.y / n, a.z / n);
}
__device__ __forceinline__ float distance3(V3 p0, V3 p1) { V3 d = p1 - p0; return sqrtf(dot3(d, d)); }

// Mat4 * (v, 1), top three rows, with nalgebra's gemv accumulation order (intersect.rs:297-301)
__device__ __forceinline__ V3 transform34(const DMat34& m, V3 v) {
    V3 r;
    r.x = ((m.r[0][0] * v.x + m.r[0][1] * v.y) + m.r[0][2] * v.z) + m.r[0][3] * 1.0f;
    r.y = ((m.r[1][0] * v.x + m.r[1][1] * v.y) + m.r[1][2] * v.z) + m.r[1][3] * 1.0f;
    r.z = ((m.r[2][0] * v.x + m.r[2][1] * v.y) + m.r[2][2] * v.z) + m.r[2][3] * 1.0f;
    return r;
}

struct Ray { V3 o, d; };

// What survives of a candidate until the closest one is known.  Position and normals are
// recomputed for the winner only (they are pure functions of the ray, the primitive and t).
struct Closest {
    bool have;
    float t;          // world-space t as the reference compares it at scene level
    uint32_t prim;    // RT_PRIM_* << 30 | index
    uint32_t inst;
    float tl, u, v;   // mesh hits: local t and barycentrics
    uint32_t slot;    // mesh hits: device triangle slot
};

// intersect.rs:367-395
struct Hitpoint {
    V3 position, hit_normal, position_for_refraction;
    bool on_frontside;
    uint32_t material;
};

__device__ __forceinline__ Hitpoint create_hitpoint(V3 hit_position, V3 ray_d, V3 surface_normal, V3 hit_normal,
                                                    uint32_t material) {
    float n_dot_rdir = dot3(surface_normal, ray_d);
    bool front = n_dot_rdir < 0.0f;
    V3 sn = front ? surface_normal : -surface_normal;
    V3 hn = front ? hit_normal : -hit_normal;
    V3 offset = sn * 1e-4f;                    // NUMERIC_ERROR_COMPENSATION_OFFSET, intersect.rs:9
    Hitpoint h;
    h.position = hit_position + offset;
    h.hit_normal = hn;
    h.position_for_refraction = hit_position - offset;
    h.on_frontside = front;
    h.material = material;
    return h;
}

// intersect.rs:114-139; returns whether there is a candidate and its t
__device__ __forceinline__ bool plane_t(const DPlane& p, const Ray& r, float& t) {
    V3 n = ld3(p.n);
    float nd = dot3(n, r.d);
    if (nd == 0.0f) return false;
    t = (p.d - dot3(n, r.o)) / nd;
    return t >= 0.0f;
}

// intersect.rs:38-112
// `a` = dot(d, d) and `dn` = normalize(d) depend on the ray only; callers with several spheres compute them once
// (same operands, same result as the reference's per-sphere evaluation).
__device__ __forceinline__ bool sphere_t(const DSphere& s, const Ray& r, float a, V3 dn, float& t) {
    V3 c = ld3(s.c);
    float radius = s.r;
    V3 m = r.o - c;
    float b = 2.0f * dot3(m, r.d);
    float cc = dot3(m, m) - radius * radius;
    V3 proj = m - dn * dot3(m, dn);
    float disc = 4.0f * a * (radius * radius - dot3(proj, proj));
    if (disc == 0.0f) {
        t = (-0.5f * b) / a;
    } else if (!(__float_as_uint(disc) >> 31)) {          // Signed::is_positive = sign bit clear
        float sq = sqrtf(disc);
        float q = (b < 0.0f) ? -0.5f * (b - sq) : -0.5f * (b + sq);
        float t0 = q / a;
        float t1 = cc / q;
        if (t0 < 0.0f && t1 >= 0.0f) t = t1;
        else if (t1 < 0.0f && t0 >= 0.0f) t = t0;
        else t = fminf(t0, t1);
    } else {
        return false;
    }
    return t >= 0.0f;
}

// intersect.rs:141-166 (Moeller-Trumbore up to the accept decision)
__device__ __forceinline__ bool triangle_tuv(V3 p0, V3 e1, V3 e2, const Ray& r, float& t, float& u, float& v) {
    V3 q = cross3(r.d, e2);
    float a = dot3(e1, q);
    if (a > -1e-5f && a < 1e-5f) return false;
    float f = 1.0f / a;
    V3 s = r.o - p0;
    u = f * dot3(s, q);
    if (u < 0.0f) return false;
    V3 rr = cross3(s, e1);
    v = f * dot3(r.d, rr);
    if (v < 0.0f || u + v > 1.0f) return false;
    t = f * dot3(e2, rr);
    if (t < 0.0f) return false;
    return true;
}

// intersect.rs:167-171: the part of Triangle::intersect after the accept decision
__device__ __forceinline__ Hitpoint triangle_hitpoint(const Ray& r, float t, float u, float v, V3 n0, V3 n1, V3 n2, V3 fn,
                                                      uint32_t material) {
    float w = 1.0f - u - v;
    V3 hit_position = r.o + r.d * t;
    V3 g = w * n0 + u * n1 + v * n2;
    g = normalize3(g);
    return create_hitpoint(hit_position, r.d, fn, g, material);
}

// utils/mod.rs:13-24
__device__ __forceinline__ bool closer(bool have, float best_t, float t) { return !have || t < best_t; }

__device__ __forceinline__ bool finite1(float x) { return (x - x) == 0.0f; }
__device__ __forceinline__ bool finite3(V3 a) { return finite1(a.x) && finite1(a.y) && finite1(a.z); }

// AABB::intersect, intersect.rs:185-226, with the ray-only reciprocal hoisted (it is recomputed per
// node there, from the same operands).  Also returns tmin.
__device__ __forceinline__ bool slab(float lx, float ly, float lz, float hx, float hy, float hz, V3 o, V3 inv, float& tmin) {
    float t1 = (lx - o.x) * inv.x;
    float t2 = (hx - o.x) * inv.x;
    float t3 = (ly - o.y) * inv.y;
    float t4 = (hy - o.y) * inv.y;
    float t5 = (lz - o.z) * inv.z;
    float t6 = (hz - o.z) * inv.z;
    tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    return !(tmax < 0.0f) && !(tmin > tmax);
}

#define RT_INF __uint_as_float(0x7F800000u)


// Pruning slack, see trace_scene: relative part of every "is this box beyond the best hit" comparison.
#define RT_PRUNE_REL 1.015625f            // 1 + 2^-6
// Leaves are tested once RT_LEAF_BATCH lanes of a wave hold one, or one walking lane in RT_LEAF_SHARE
// (measured on the 1080p bunnies frame: 16/4 2.02 ms, 32/2 1.71 ms, 48/2 1.72 ms of BVH walk per frame).
#ifndef RT_LEAF_BATCH
#define RT_LEAF_BATCH 32
#endif
#ifndef RT_REFILL_IDLE
#define RT_REFILL_IDLE 32          // idle lanes of a wave that make it fetch new pairs (with 4-wide nodes: 16: 1.243, 24: 1.234,
                                   // 32: 1.222, 40: 1.28 ms of BVH walk per 1080p bunnies frame)
#endif
#ifndef RT_SHARE_IDLE
#define RT_SHARE_IDLE 16           // idle lanes that make a wave without input share its pending subtrees
#endif
#ifndef RT_LEAF_SHARE
#define RT_LEAF_SHARE 2
#endif
#ifndef RT_CONTIGUOUS_MIN_CHUNKS
#define RT_CONTIGUOUS_MIN_CHUNKS 4   // chunks of 64 pairs per wave from which a launch of the BVH walk hands out runs of neighbouring pairs
#endif

// Full hitpoint of the winner.
__device__ __forceinline__ Hitpoint finalize_hit(const DScene& S, const Ray& r, const Closest& c) {
    uint32_t kind = c.prim >> 30, idx = c.prim & 0x3FFFFFFFu;
    if (kind == RT_PRIM_PLANE) {
        const DPlane& p = S.planes[idx];
        V3 n = ld3(p.n);
        return create_hitpoint(r.o + r.d * c.t, r.d, n, n, p.material);
    } else if (kind == RT_PRIM_SPHERE) {
        const DSphere& s = S.spheres[idx];
        V3 hp = r.o + r.d * c.t;
        V3 n = normalize3(hp - ld3(s.c));                      // sphere.rs:12-15
        return create_hitpoint(hp, r.d, n, n, s.material);
    } else if (kind == RT_PRIM_TRIANGLE) {
        const DTriangle& tr = S.triangles[idx];
        float t, u, v;
        triangle_tuv(ld3(tr.p0), ld3(tr.e1), ld3(tr.e2), r, t, u, v);
        return triangle_hitpoint(r, t, u, v, ld3(tr.n0), ld3(tr.n1), ld3(tr.n2), ld3(tr.fn), tr.material);
    } else {
        const DInstance& inst = S.instances[c.inst];
        Ray lr;
        lr.o = transform34(inst.model_inverse, r.o);
        lr.d = normalize3(transform34(inst.rotation_scale_inverse, r.d));
        float4 h2 = S.tri_hot[3 * c.slot + 2];
        float4 c0 = S.tri_cold[3 * c.slot], c1 = S.tri_cold[3 * c.slot + 1], c2 = S.tri_cold[3 * c.slot + 2];
        Hitpoint h = triangle_hitpoint(lr, c.tl, c.u, c.v, mk(c0.x, c0.y, c0.z), mk(c1.x, c1.y, c1.z), mk(c2.x, c2.y, c2.z),
                                       mk(h2.y, h2.z, h2.w), __float_as_uint(c1.w));
        h.position = transform34(inst.model, h.position);
        h.hit_normal = normalize3(transform34(inst.rotation_scale, h.hit_normal));
        h.position_for_refraction = transform34(inst.model, h.position_for_refraction);
        if (inst.material_override >= 0) h.material = (uint32_t)inst.material_override;
        return h;
    }
}

// raytracer.rs:53-63
__device__ __forceinline__ V3 background(const DScene& S, V3 dir) {
    if (S.background.type == RT_BACKGROUND_SOLID_COLOR) return ld3(S.background.color);
    return (dir + mk(1.0f, 1.0f, 1.0f)) * 0.5f;
}

// raytracer.rs:190-195
__device__ __forceinline__ V3 reflected_dir(V3 V, V3 N) { return 2.0f * dot3(N, V) * N - V; }

// raytracer.rs:197-207
__device__ __forceinline__ V3 transmitted_dir(V3 L, V3 N, float n1, float n2) {
    float n = n1 / n2;
    float w = n * dot3(L, N);
    float k = sqrtf(1.0f + (w - n) * (w + n));
    return (w - k) * N - n * L;
}

// raytracer.rs:222-240
__device__ __forceinline__ float fresnel_reflection(V3 refl_d, V3 trans_d, V3 N, float n1, float n2) {
    V3 TN = -N;
    float cos_i = dot3(refl_d, N);
    float cos_t = dot3(trans_d, TN);
    float r_par = (n2 * cos_i - n1 * cos_t) / (n2 * cos_i + n1 * cos_t);
    float r_ort = (n1 * cos_i - n2 * cos_t) / (n1 * cos_i + n2 * cos_t);
    return 0.5f * (r_par * r_par + r_ort * r_ort);
}

// ---- pixel sums --------------------------------------------------------------------------------
// Contributions below 2^22 in magnitude go to a sum in 2^-40 units (exact for every float >= 2^-16, and far
// finer than the 8-bit result needs below that): integer sums, so the result does not depend on the order in
// which rays finish.  Larger finite ones - Fresnel factors can blow up, and multiply along a path, where a shading
// normal disagrees with the geometry - go to a second sum in double precision, whose rounding is far below what
// such a pixel's float arithmetic in the reference resolves.
// poison bits per channel c: bit c = NaN, bit 3+c = +inf, bit 6+c = -inf; bit 9: the second sum was used.
#define RT_FIXED_ONE 1099511627776.0     // 2^40
#define RT_FIXED_LIMIT 4194304.0f        // 2^22
#define RT_POISON_BIG 512u

__device__ __forceinline__ void add_channel(const DFrame& F, uint32_t pixel, int c, float v) {
    if (v == 0.0f) return;
    if (fabsf(v) < RT_FIXED_LIMIT) {
        long long q = __double2ll_rn((double)v * RT_FIXED_ONE);
#if defined(RT_PROBE_PLAIN_COLOUR_STORES)
        F.acc[(size_t)c * F.n_pixels + pixel] = q;                  // timing probe only: wrong pictures
#elif defined(RT_PROBE_NO_COLOUR)
        if (q == 0x7FFFFFFFFFFFFFFFll) F.acc[(size_t)c * F.n_pixels + pixel] = q;      // timing probe only: (never true)
#else
        atomicAdd(reinterpret_cast<unsigned long long*>(F.acc + (size_t)c * F.n_pixels + pixel), (unsigned long long)q);
#endif
    } else if (finite1(v)) {
        atomicAdd(F.acc_big + (size_t)c * F.n_pixels + pixel, (double)v);
        atomicOr(F.poison + pixel, RT_POISON_BIG);
    } else {
        unsigned int bit = (v != v) ? (1u << c) : (v > 0.0f ? (8u << c) : (64u << c));
        atomicOr(F.poison + pixel, bit);
    }
}

__device__ __forceinline__ void add_color(const DFrame& F, uint32_t pixel, V3 c) {
    add_channel(F, pixel, 0, c.x);
    add_channel(F, pixel, 1, c.y);
    add_channel(F, pixel, 2, c.z);
}

// ---- ray records -------------------------------------------------------------------------------
#define RT_META_PIXEL_MASK 0x0FFFFFFFu     // RT_MAX_PIXELS_PER_SHARD - 1
#define RT_META_KIND_SHIFT 28            // 0 = primary / reflected, 1 = transmitted

#define RT_RAY_SHADOW_BIT 0x80000000u    // = RT_RAY_SHADOW, see the work lists below
struct RayRec { Ray ray; float throughput; uint32_t meta; };

__device__ __forceinline__ void store_rec(float4* q, uint32_t i, const RayRec& rr) {
    q[2 * (size_t)i] = make_float4(rr.ray.o.x, rr.ray.o.y, rr.ray.o.z, rr.ray.d.x);
    q[2 * (size_t)i + 1] = make_float4(rr.ray.d.y, rr.ray.d.z, rr.throughput, __uint_as_float(rr.meta));
}
__device__ __forceinline__ RayRec load_rec(const float4* q, uint32_t i) {
    float4 a = q[2 * (size_t)i], b = q[2 * (size_t)i + 1];
    RayRec rr;
    rr.ray.o = mk(a.x, a.y, a.z);
    rr.ray.d = mk(a.w, b.x, b.y);
    rr.throughput = b.z;
    rr.meta = __float_as_uint(b.w);
    return rr;
}

// Shadow rays are not stored: the shadow ray of hit record `hs` towards light `li` (ray slot shadow_base + hs * n_lights
// + li) is rebuilt from the record wherever it is needed - trace_shadow_ray's own arithmetic (raytracer.rs:158-167) on the
// stored hit position, so the same bits every time.
__device__ __forceinline__ RayRec shadow_rec_of(const DScene& S, const float4 h0, uint32_t li) {
    const DLight& L = S.lights[li];
    const V3 position = mk(h0.x, h0.y, h0.z);
    const V3 dir = (L.pos[3] == 0.0f) ? mk(L.pos[0], L.pos[1], L.pos[2]) : ld3(L.world) - position;
    RayRec rr;
    rr.ray.o = position;
    rr.ray.d = normalize3(dir);
    rr.throughput = 0.0f;
    rr.meta = RT_RAY_SHADOW_BIT | li;
    return rr;
}
__device__ __forceinline__ RayRec shadow_rec(const DScene& S, const float4* hits, uint32_t hs, uint32_t li) {
    return shadow_rec_of(S, hits[3 * (size_t)hs], li);
}

// The two cosines of radiance() (raytracer.rs:139-145) for the unit vector l towards the light: the one place they are
// computed, so that whoever asks gets the same bits.
__device__ __forceinline__ void phong_cosines(V3 l, V3 n, V3 view, float& l_dot_n, float& r_dot_v) {
    const V3 rv = reflected_dir(l, n);
    l_dot_n = fmaxf(dot3(l, n), 0.0f);
    r_dot_v = fmaxf(dot3(rv, view), 0.0f);
}

// A shadow ray's verdict only ever switches the diffuse and the specular term of radiance() off (raytracer.rs:149-150).
// Where both are zero anyway - the surface faces away from the light (l.n <= 0, so the product with the clamped cosine is
// 0) or the colour product is zero, and likewise the mirrored light direction faces away from the viewer (0 to a positive
// power) - tracing the ray cannot change a bit of the picture, and it is not traced: its key stays "no hit".  The
// conditions are on the very floats the Phong sum uses later (phong_cosines on the stored hit record), and the products
// must not turn 0 * x into NaN, hence the finiteness flags.  Shadow rays of primary hits are traced regardless when the
// caller wants the per-pixel flags, which report the verdict itself (RT_FLAG_SHADOW_LIGHT0).
__device__ __forceinline__ bool shadow_is_moot(const DScene& S, const DFrame& F, const float4 h0, const float4 h1, const float4 h2, uint32_t li, V3 l) {
    const uint32_t flags = S.moot_flags[__float_as_uint(h0.w) * S.n_lights + li];
    if (!flags) return false;
    if (F.aux_flags && (__float_as_uint(h2.w) & 0x80000000u)) return false;       // RT_HIT_DEPTH0
    float l_dot_n, r_dot_v;
    phong_cosines(l, mk(h1.x, h1.y, h1.z), mk(h2.x, h2.y, h2.z), l_dot_n, r_dot_v);
    const bool diffuse_fixed = ((flags & RT_MOOT_DIFFUSE_ZERO) && l_dot_n < RT_INF) || ((flags & RT_MOOT_DIFFUSE_FINITE) && l_dot_n == 0.0f);
    const bool specular_fixed = ((flags & RT_MOOT_SPECULAR_ZERO) && r_dot_v <= 1.0f) || ((flags & RT_MOOT_SPECULAR_FINITE) && r_dot_v == 0.0f);
    return diffuse_fixed && specular_fixed;
}
// Any ray of recursion depth `level` by its slot: listed rays from the ray list, shadow rays from the hits of the depth above.
__device__ __forceinline__ RayRec load_any_rec(const DScene& S, const DFrame& F, uint32_t level, uint32_t gi) {
    if (gi < F.shadow_base) return load_rec(F.rays[level & 1], gi);
    const uint32_t k = gi - F.shadow_base, hs = k / S.n_lights;
    return shadow_rec(S, F.hits[(level & 1) ^ 1], hs, k - hs * S.n_lights);
}
__device__ __forceinline__ V3 load_any_origin(const DScene& S, const DFrame& F, uint32_t level, uint32_t gi) {
    const float4 a = gi < F.shadow_base ? F.rays[level & 1][2 * (size_t)gi] : F.hits[(level & 1) ^ 1][3 * (size_t)((gi - F.shadow_base) / S.n_lights)];
    return mk(a.x, a.y, a.z);
}

// local pixel index -> view and canvas (col, row); false for padding pixels outside the frame
__device__ __forceinline__ bool pixel_coords(const DFrame& F, uint32_t p, uint32_t& view, uint32_t& col, uint32_t& row) {
    view = p / F.view_pixels;
    p -= view * F.view_pixels;
    uint32_t k = p / (RT_TILE_DIM * RT_TILE_DIM), q = p % (RT_TILE_DIM * RT_TILE_DIM);
    uint32_t tile = F.shard_index + k * F.shard_count;
    if (tile >= F.tiles_x * F.tiles_y) return false;
    uint32_t b = q / 64, l = q % 64;                       // 8x8 pixel blocks, one per wave
    uint32_t px = (b % 4) * 8 + (l % 8), py = (b / 4) * 8 + (l / 8);
    col = (tile % F.tiles_x) * RT_TILE_DIM + px;
    row = (tile / F.tiles_x) * RT_TILE_DIM + py;
    return col < F.width && row < F.height;
}

// raytracer.rs:30-51 for canvas pixel (col, row); screen y = H-1-row (canvas.rs:35-46)
__device__ __forceinline__ Ray primary_ray(const DFrame& F, const float* m, uint32_t col, uint32_t row) {
    float x = (float)col, y = (float)(F.height - 1 - row);
    float p[4] = {x, y, 0.0f, 1.0f};
    float pf[4] = {x + 0.0f, y + 0.0f, 0.0f + 1.0f, 1.0f + 0.0f};
    float pw[4], pfw[4];
    for (int i = 0; i < 4; ++i) {
        pw[i] = ((m[i] * p[0] + m[4 + i] * p[1]) + m[8 + i] * p[2]) + m[12 + i] * p[3];
        pfw[i] = ((m[i] * pf[0] + m[4 + i] * pf[1]) + m[8 + i] * pf[2]) + m[12 + i] * pf[3];
    }
    V3 a = mk(pw[0] / pw[3], pw[1] / pw[3], pw[2] / pw[3]);
    V3 b = mk(pfw[0] / pfw[3], pfw[1] / pfw[3], pfw[2] / pfw[3]);
    Ray r;
    r.o = a;
    r.d = normalize3(b - a);
    return r;
}


// ---- work lists ------------------------------------------------------------------------------------

#define RT_KEY_NONE 0xFFFFFFFFFFFFFFFFull
#define RT_KEY_MESH 0x80000000u          // low word: bit 31 set = BVH work item index, clear = analytic primitive
#define RT_RAY_SHADOW 0x80000000u        // ray meta: shadow ray; low 8 bits = light index, record.w = bits(hit slot)
#define RT_HIT_DEPTH0 0x80000000u        // hit record pixel word: the hit belongs to a primary ray

// Candidate order as one integer: smaller t first; at equal t the candidate the reference meets first
// (planes, spheres, triangles, then mesh instances in Vec order: scene.rs:129-136 with the strict `<`
// of utils/mod.rs:17).  Only used for rays whose components are all finite: their t are finite and
// >= 0, and the bit pattern of a non-negative float grows with its value.
__device__ __forceinline__ unsigned long long make_key(float t, uint32_t order) {
    uint32_t tb = (t == 0.0f) ? 0u : __float_as_uint(t);          // -0.0 compares equal to +0.0
    return ((unsigned long long)tb << 32) | order;
}
__device__ __forceinline__ uint32_t analytic_order(uint32_t kind, uint32_t index) { return (kind << 29) | index; }

// Exclusive prefix of the segment counts into `pre[0..n_seg]` (LDS); `scratch` is blockDim.x words of
// LDS.  Block-wide: every thread of the block must call it.  Returns the total.
// Each thread owns a run of consecutive segments; the runs are scanned inside each wave with lane shifts
// and across the waves through `scratch`, so the whole thing costs two barriers.
#define RT_PREFIX_RUN 8            // longest run a thread keeps in registers: 256 threads x 8 = 2048 segments
__device__ uint32_t build_prefix(const unsigned int* counts, uint32_t n_seg, uint32_t seg_cap, uint32_t ovf_cap,
                                 uint32_t* pre, uint32_t* scratch) {
    const uint32_t t = threadIdx.x, nt = blockDim.x;
    const uint32_t lane = t & (RT_WAVE - 1), wave = t / RT_WAVE, n_wave = nt / RT_WAVE;
    const uint32_t per = (n_seg + nt - 1) / nt;
    const uint32_t b0 = t * per;
    uint32_t sum = 0;
    uint32_t mine[RT_PREFIX_RUN];
    if (per <= RT_PREFIX_RUN) {
#pragma unroll
        for (uint32_t k = 0; k < RT_PREFIX_RUN; ++k) {
            const uint32_t i = b0 + k;
            uint32_t c = 0;
            if (k < per && i < n_seg) { c = counts[i]; const uint32_t cap = (i + 1 == n_seg) ? ovf_cap : seg_cap; c = c < cap ? c : cap; }
            mine[k] = c;
            sum += c;
        }
    } else {
        for (uint32_t i = b0; i < b0 + per && i < n_seg; ++i) {
            uint32_t c = counts[i], cap = (i + 1 == n_seg) ? ovf_cap : seg_cap;
            sum += c < cap ? c : cap;
        }
    }
    // inclusive scan of `sum` over the lanes of the wave
    uint32_t incl = sum;
#pragma unroll
    for (uint32_t off = 1; off < RT_WAVE; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == RT_WAVE - 1) scratch[wave] = incl;
    __syncthreads();
    uint32_t run = incl - sum, total = 0;
    for (uint32_t w = 0; w < n_wave; ++w) {
        const uint32_t v = scratch[w];
        if (w < wave) run += v;
        total += v;
    }
    if (per <= RT_PREFIX_RUN) {
#pragma unroll
        for (uint32_t k = 0; k < RT_PREFIX_RUN; ++k) {
            const uint32_t i = b0 + k;
            if (k < per && i < n_seg) { pre[i] = run; run += mine[k]; }
        }
    } else {
        for (uint32_t i = b0; i < b0 + per && i < n_seg; ++i) {
            uint32_t c = counts[i], cap = (i + 1 == n_seg) ? ovf_cap : seg_cap;
            pre[i] = run;
            run += c < cap ? c : cap;
        }
    }
    if (t == 0) pre[n_seg] = total;
    __syncthreads();
    return total;
}

// Element v of the concatenated segments -> its slot in the backing array.
__device__ __forceinline__ uint32_t locate(const uint32_t* pre, uint32_t n_seg, uint32_t seg_cap, uint32_t v) {
    uint32_t lo = 0, hi = n_seg;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (pre[mid] <= v) lo = mid; else hi = mid;
    }
    return lo * seg_cap + (v - pre[lo]);          // the overflow segment starts at (n_seg - 1) * seg_cap
}

// The same for the 64 consecutive elements v0 .. v0 + 63 a wave handles together (lane l gets element v0 + l;
// must be called by all lanes of the wave; elements at or beyond `total` get an unspecified slot).  The
// wave finds the segment of v0 in two cooperative steps - 64 samples of the table, then the bracket between
// two samples - instead of ten dependent ones, and each lane walks on from there.
// `off`: which of the consecutive elements v0, v0 + 1, ... this lane wants (lanes with !wanted get an unspecified slot).
__device__ __forceinline__ uint32_t locate_run(const uint32_t* pre, uint32_t n_seg, uint32_t seg_cap, uint32_t v0, uint32_t off, bool wanted,
                                               uint32_t lane) {
    uint32_t lo = 0, span = n_seg + 1;                     // the answer for v0 lies in [lo, lo + span)
    while (span > 1) {
        const uint32_t stride = (span + RT_WAVE - 1) / RT_WAVE;
        const uint32_t i = lo + lane * stride;
        const bool below = lane * stride < span && pre[i] <= v0;
        const uint32_t cnt = __popcll(__ballot(below));    // pre[lo] <= v0 always, so cnt >= 1
        lo += (cnt - 1) * stride;
        span = (lo + stride <= n_seg + 1) ? stride : n_seg + 1 - lo;
    }
    uint32_t seg = lo < n_seg ? lo : n_seg - 1;
    const uint32_t v = v0 + off;
    if (wanted) {
        // pre[n_seg] = total > v ends the walk; a long run of empty segments is crossed by bisection
        uint32_t steps = 0;
        while (pre[seg + 1] <= v) {
            ++seg;
            if (++steps == 4) {
                uint32_t a = seg, b = n_seg;
                while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (pre[mid] <= v) a = mid; else b = mid; }
                seg = a;
                break;
            }
        }
    }
    return seg * seg_cap + (v - pre[seg]);
}
__device__ __forceinline__ uint32_t locate_chunk(const uint32_t* pre, uint32_t n_seg, uint32_t seg_cap, uint32_t v0, uint32_t total,
                                                 uint32_t lane) {
    return locate_run(pre, n_seg, seg_cap, v0, lane, v0 + lane < total, lane);
}

// A slot of this block's output segment for every lane that wants one (RT_REF_NONE otherwise, or when
// even the overflow segment is full).  Must be called by all lanes of the wave.
__device__ __forceinline__ uint32_t append_slot(bool want, uint32_t* lds_counter, uint32_t seg_cap, uint32_t n_seg,
                                                unsigned int* counts, uint32_t ovf_cap, unsigned int* overflow_flag,
                                                uint32_t overflow_seq, uint32_t lane) {
    unsigned long long mask = __ballot(want);
    if (!mask) return RT_REF_NONE;
    uint32_t leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(lds_counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (!want) return RT_REF_NONE;
    uint32_t k = base + __popcll(mask & ((1ull << lane) - 1ull));
    if (k < seg_cap) return blockIdx.x * seg_cap + k;
    uint32_t j = atomicAdd(&counts[n_seg - 1], 1u);
    if (j < ovf_cap) return (n_seg - 1) * seg_cap + j;
    *overflow_flag = overflow_seq;
    return RT_REF_NONE;
}

// The lists of one recursion depth (see DFrame::counts).
struct LevelLists {
    unsigned int* ray_counts;        // rays traced at this depth
    unsigned int* hit_counts;        // hit records of the depth above
    unsigned int* fat_counts;
    unsigned int* slow_counts;
    unsigned int* next_ray_counts;   // what rt_shade appends to at this depth
    unsigned int* next_hit_counts;
    unsigned long long* best;
    float4* fat_items;
    float4* item_results;
    unsigned int* heavy_count;       // pairs left to traverse_heavy_pairs at this depth
    unsigned int* heavy_items;
};
__device__ __forceinline__ LevelLists level_lists(const DFrame& F, uint32_t level) {
    LevelLists L;
    unsigned int* row = F.counts + (size_t)RT_COUNT_ROWS * level * F.n_seg;
    L.ray_counts = row + RT_ROW_RAYS * F.n_seg;
    L.hit_counts = row + RT_ROW_HITS * F.n_seg;
    L.fat_counts = row + RT_ROW_FAT * F.n_seg;
    L.slow_counts = row + RT_ROW_SLOW * F.n_seg;
    L.next_ray_counts = L.ray_counts + RT_COUNT_ROWS * F.n_seg;
    L.next_hit_counts = L.hit_counts + RT_COUNT_ROWS * F.n_seg;
    L.best = F.best[level & 1];
    L.fat_items = F.fat_items[level & 1];
    L.item_results = F.item_results[level & 1];
    L.heavy_count = F.heavy_count + level;
    L.heavy_items = F.heavy_items + (size_t)level * RT_HEAVY_CAP;
    return L;
}

// ---- exact walk for rays with a non-finite component ------------------------------------------------
// intersect.rs:228-269 in the reference's own order, no pruning.  A candidate with t = NaN ends the walk:
// nothing can replace it (utils/mod.rs:17).
__device__ void exact_mesh_hit(const DScene& S, uint32_t root_ref, const Ray& lr, bool& m_have, float& m_t, float& m_u,
                               float& m_v, uint32_t& m_slot) {
    uint32_t stk[RT_BVH_STACK_LEN];
    V3 inv = mk(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
    m_have = false; m_t = 0.0f; m_u = 0.0f; m_v = 0.0f; m_slot = 0;
    uint32_t sp = 0;
    stk[sp++] = root_ref;
    while (sp > 0) {
        const uint32_t ref = stk[--sp];
        if (ref & RT_REF_LEAF_BIT) {
            const uint32_t first = ref & RT_REF_SLOT_MASK, cnt = (ref >> RT_REF_COUNT_SHIFT) & 7u;
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t slot = first + k;
                float4 h0 = S.tri_hot[3 * slot], h1 = S.tri_hot[3 * slot + 1], h2 = S.tri_hot[3 * slot + 2];
                float t, u, v;
                if (triangle_tuv(mk(h0.x, h0.y, h0.z), mk(h0.w, h1.x, h1.y), mk(h1.z, h1.w, h2.x), lr, t, u, v) &&
                    closer(m_have, m_t, t)) {
                    m_have = true; m_t = t; m_u = u; m_v = v; m_slot = slot;
                }
            }
            if (m_have && m_t != m_t) return;
        } else {
            const float4 n0 = S.nodes[4 * ref], n1 = S.nodes[4 * ref + 1], n2 = S.nodes[4 * ref + 2], n3 = S.nodes[4 * ref + 3];
            float tl, tr;
            bool hl = slab(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, lr.o, inv, tl);
            bool hr = slab(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, lr.o, inv, tr);
            if (hl && sp < RT_BVH_STACK_LEN) stk[sp++] = __float_as_uint(n3.x);      // left first: right is popped first
            if (hr && sp < RT_BVH_STACK_LEN) stk[sp++] = __float_as_uint(n3.y);
        }
    }
}

// world-space t of an instance's closest local hit: intersect.rs:312-319
__device__ __forceinline__ float world_t(const DScene& S, const DInstance& inst, const Ray& lr, V3 world_origin, float m_t,
                                         uint32_t m_slot) {
    float4 h2 = S.tri_hot[3 * m_slot + 2];
    V3 fn = mk(h2.y, h2.z, h2.w);
    V3 p = lr.o + lr.d * m_t;
    V3 sn = (dot3(fn, lr.d) < 0.0f) ? fn : -fn;
    return distance3(world_origin, transform34(inst.model, p + sn * 1e-4f));
}

struct ShadowSpec { bool is_shadow, directional; float limit_world; };

__device__ __forceinline__ ShadowSpec shadow_spec(const DScene& S, const RayRec& rr) {
    ShadowSpec s;
    s.is_shadow = (rr.meta & RT_RAY_SHADOW) != 0;
    s.directional = false;
    s.limit_world = RT_INF;
    if (s.is_shadow) {
        const DLight& L = S.lights[rr.meta & 0xFFu];
        s.directional = L.pos[3] == 0.0f;
        if (!s.directional) s.limit_world = distance3(rr.ray.o, ld3(L.world));     // raytracer.rs:177-178
    }
    return s;
}

// The 64-byte walk item of a (ray, instance) pair (see RT_FAT_FAST in rt_device.h): the mesh-local ray of
// intersect.rs:296-308 and the test of the mesh's root box (the first iteration of intersect.rs:245-250).
// `lw`: world distance beyond which nothing on this ray matters any more (closest analytic hit, light of a shadow
// ray; +inf if none).  Returns false when the pair needs no walk.
__device__ __forceinline__ bool walk_item(const DInstance& inst, uint32_t ii, uint32_t gi, const Ray& r, const ShadowSpec& ss, float lw,
                                          float4& q0, float4& q1, float4& q2, float4& q3) {
    const float o_l1 = fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z);
    Ray lr;
    lr.o = transform34(inst.model_inverse, r.o);
    const V3 dv = transform34(inst.rotation_scale_inverse, r.d);
    const float dn = sqrtf(dot3(dv, dv));
    lr.d = mk(dv.x / dn, dv.y / dn, dv.z / dn);                 // glm::normalize, intersect.rs:304-308
    const V3 inv = mk(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
    const bool fast = inst.prunable != 0;
    const float slack_w = 2.0f * inst.world_guard;
    // mesh-local bound on useful t: a hit at world distance w lies at local t = w * dn (up to world_guard)
    float limit = RT_INF;
    if (fast && lw < RT_INF) limit = (lw + slack_w + 1.52587890625e-05f * (o_l1 + lw)) * dn * 1.0009765625f;
    float root_tmin;
    bool pass = slab(inst.root_lo[0], inst.root_lo[1], inst.root_lo[2], inst.root_hi[0], inst.root_hi[1], inst.root_hi[2],
                     lr.o, inv, root_tmin);
    if (pass && fast && root_tmin > limit * RT_PRUNE_REL + inst.prune_abs) pass = false;
    // Shadow rays: a candidate at local t lies within world_guard of world distance t / dn, and the pair's
    // final hit is not farther than any candidate.
    float thr = -RT_INF;
    if (ss.is_shadow) thr = ss.directional ? RT_INF : ss.limit_world - (1.5f * slack_w + 6.103515625e-05f * (o_l1 + ss.limit_world));
    q0 = make_float4(lr.o.x, lr.o.y, lr.o.z, root_tmin);
    q1 = make_float4(lr.d.x, lr.d.y, lr.d.z, limit);
    q2 = make_float4(1.0f / dn, thr, inst.prune_abs, __uint_as_float(inst.root_ref4));
    // A ray whose local origin lies beyond heavy_origin on two axes or more has lost the boxes on those axes - its slab
    // products there are all the same few values - and passes a tube's worth of leaves: a candidate for
    // traverse_heavy_pairs, see store_walk_item.  (With one such axis the other two still cut the tree down.)
    const uint32_t far_axes = (fabsf(lr.o.x) > inst.heavy_origin ? 1u : 0u) + (fabsf(lr.o.y) > inst.heavy_origin ? 1u : 0u) +
                              (fabsf(lr.o.z) > inst.heavy_origin ? 1u : 0u);
    const bool heavy = pass && far_axes >= 2u;
    q3 = make_float4(__uint_as_float(gi), __uint_as_float(ii | (fast ? RT_FAT_FAST : 0u) | (heavy ? RT_FAT_HEAVY : 0u)), 0.0f, 0.0f);
    return pass;
}

// Stores a walk item.  One that walk_item marked heavy goes on the depth's list for traverse_heavy_pairs as well - while that
// list has room; otherwise it loses the mark and is walked like any other (the mark only chooses who computes the pair).
__device__ __forceinline__ void store_walk_item(const LevelLists& L, uint32_t slot, float4 q0, float4 q1, float4 q2, float4 q3) {
    if (__float_as_uint(q3.y) & RT_FAT_HEAVY) {
        const uint32_t k = atomicAdd(L.heavy_count, 1u);
        if (k < RT_HEAVY_CAP) L.heavy_items[k] = slot;
        else q3.y = __uint_as_float(__float_as_uint(q3.y) & ~RT_FAT_HEAVY);
    }
    float4* fat_items = L.fat_items;
    fat_items[4 * (size_t)slot] = q0; fat_items[4 * (size_t)slot + 1] = q1;
    fat_items[4 * (size_t)slot + 2] = q2; fat_items[4 * (size_t)slot + 3] = q3;
}

// ---- the pairs of a wave waiting for their walk items (rt_setup) ----------------------------------------------------
// A ray passes the culling boxes of some instances only, so the lanes that have a pair for a given instance are few; the
// local ray and the root test are therefore not computed in place but for 64 pairs at a time: the wave keeps its pairs
// in a ring in LDS (first in, first out: a ray's pairs stay in the instances' Vec order, which the slots of the walk
// items must keep, see RT_KEY_MESH), and whenever 64 are waiting one lane takes one pair each.
#define RT_RING_CAP 128                   // at most 63 pairs are left over when at most 64 more arrive
#define RT_RING_WORDS (9 * RT_RING_CAP)   // word w of entry e: ring[w * RT_RING_CAP + e]: origin, direction, lw, ray slot, instance + flags
#define RT_PAIR_LIGHT_SHIFT 20            // = log2(RT_MAX_INSTANCES)
#define RT_PAIR_SHADOW 0x10000000u
#define RT_PAIR_DIRECTIONAL 0x20000000u

struct PairRing { uint32_t* ring; uint32_t head, count; };

__device__ __forceinline__ void ring_push(PairRing& R, bool want, const Ray& r, float lw, uint32_t gi, uint32_t ii, const RayRec& rr,
                                          const ShadowSpec& ss, uint32_t lane) {
    const unsigned long long mask = __ballot(want);
    if (want) {
        const uint32_t e = (R.head + R.count + __popcll(mask & ((1ull << lane) - 1ull))) & (RT_RING_CAP - 1);
        uint32_t* q = R.ring + e;
        q[0 * RT_RING_CAP] = __float_as_uint(r.o.x); q[1 * RT_RING_CAP] = __float_as_uint(r.o.y); q[2 * RT_RING_CAP] = __float_as_uint(r.o.z);
        q[3 * RT_RING_CAP] = __float_as_uint(r.d.x); q[4 * RT_RING_CAP] = __float_as_uint(r.d.y); q[5 * RT_RING_CAP] = __float_as_uint(r.d.z);
        q[6 * RT_RING_CAP] = __float_as_uint(lw);
        q[7 * RT_RING_CAP] = gi;
        q[8 * RT_RING_CAP] = ii | (ss.is_shadow ? (RT_PAIR_SHADOW | ((rr.meta & 0xFFu) << RT_PAIR_LIGHT_SHIFT)) : 0u) |
                             (ss.directional ? RT_PAIR_DIRECTIONAL : 0u);
    }
    R.count += (uint32_t)__popcll(mask);
    __builtin_amdgcn_wave_barrier();
}

// Turns up to 64 waiting pairs into walk items.  Must be called by all lanes of the wave.
__device__ __forceinline__ void ring_drain(PairRing& R, const DScene& S, const DFrame& F, const LevelLists& L, uint32_t* out_n, uint32_t lane) {
    const uint32_t n = R.count < RT_WAVE ? R.count : RT_WAVE;
    bool pass = false;
    float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
    if (lane < n) {
        const uint32_t* q = R.ring + ((R.head + lane) & (RT_RING_CAP - 1));
        Ray r;
        r.o = mk(__uint_as_float(q[0 * RT_RING_CAP]), __uint_as_float(q[1 * RT_RING_CAP]), __uint_as_float(q[2 * RT_RING_CAP]));
        r.d = mk(__uint_as_float(q[3 * RT_RING_CAP]), __uint_as_float(q[4 * RT_RING_CAP]), __uint_as_float(q[5 * RT_RING_CAP]));
        const float lw = __uint_as_float(q[6 * RT_RING_CAP]);
        const uint32_t gi = q[7 * RT_RING_CAP], w = q[8 * RT_RING_CAP], ii = w & (RT_MAX_INSTANCES - 1u);
        ShadowSpec ss;
        ss.is_shadow = (w & RT_PAIR_SHADOW) != 0;
        ss.directional = (w & RT_PAIR_DIRECTIONAL) != 0;
        ss.limit_world = RT_INF;
        if (ss.is_shadow && !ss.directional) ss.limit_world = distance3(r.o, ld3(S.lights[(w >> RT_PAIR_LIGHT_SHIFT) & 0xFFu].world));   // as shadow_spec
        pass = walk_item(S.instances[ii], ii, gi, r, ss, lw, q0, q1, q2, q3);
    }
    __builtin_amdgcn_wave_barrier();
    R.head = (R.head + n) & (RT_RING_CAP - 1);
    R.count -= n;
    const uint32_t slot = append_slot(pass, out_n, F.seg_fat, F.n_seg, L.fat_counts, F.ovf_fat, F.overflow, F.overflow_seq, lane);
    if (slot != RT_REF_NONE) store_walk_item(L, slot, q0, q1, q2, q3);
}

// Scene::intersect's three analytic groups (scene.rs:129-131 over intersect.rs:20-36): per group the closest
// hit in Vec order, groups merged with the strict `<` of utils/mod.rs:17.  `settled`: a shadow ray that is
// already known to be blocked needs nothing further (raytracer.rs:171-187).
__device__ __forceinline__ void analytic_hits(const DScene& S, const DPlane* planes, const DSphere* spheres,
                                              const DTriangle* triangles, const Ray& r, bool early, const ShadowSpec& ss,
                                              bool& have, float& best_t, uint32_t& best_order, bool& settled) {
    {
        bool g_have = false; float g_t = 0.0f; uint32_t g_i = 0;
        for (uint32_t i = 0; i < S.n_planes; ++i) {
            float t;
            if (plane_t(planes[i], r, t) && closer(g_have, g_t, t)) { g_have = true; g_t = t; g_i = i; }
        }
        if (g_have && closer(have, best_t, g_t)) { have = true; best_t = g_t; best_order = analytic_order(RT_PRIM_PLANE, g_i); }
    }
    settled = early && have && (ss.directional || best_t < ss.limit_world);
    if (!settled) {
        bool g_have = false; float g_t = 0.0f; uint32_t g_i = 0;
        const float dd = dot3(r.d, r.d);
        const V3 dn = normalize3(r.d);
        for (uint32_t i = 0; i < S.n_spheres; ++i) {
            float t;
            if (sphere_t(spheres[i], r, dd, dn, t) && closer(g_have, g_t, t)) { g_have = true; g_t = t; g_i = i; }
        }
        if (g_have && closer(have, best_t, g_t)) { have = true; best_t = g_t; best_order = analytic_order(RT_PRIM_SPHERE, g_i); }
        settled = early && have && (ss.directional || best_t < ss.limit_world);
    }
    if (!settled) {
        bool g_have = false; float g_t = 0.0f; uint32_t g_i = 0;
        for (uint32_t i = 0; i < S.n_triangles; ++i) {
            const DTriangle& tr = triangles[i];
            float t, u, w;
            if (triangle_tuv(ld3(tr.p0), ld3(tr.e1), ld3(tr.e2), r, t, u, w) && closer(g_have, g_t, t)) {
                g_have = true; g_t = t; g_i = i;
            }
        }
        if (g_have && closer(have, best_t, g_t)) { have = true; best_t = g_t; best_order = analytic_order(RT_PRIM_TRIANGLE, g_i); }
        settled = early && have && (ss.directional || best_t < ss.limit_world);
    }
}

// Words of LDS the analytic primitives take when staged (rt_setup), 0 when they are too many to stage.  Behind them, in
// scenes whose instances are scanned one by one (no range tree: at most 16 by default, 64 at most), a culling record of RT_CULL_WORDS words per
// instance: world_lo, world_hi, cull_origin_limit, bits root_ref.
#define RT_STAGE_WORDS_MAX 4096
#define RT_CULL_WORDS 8
__host__ __device__ inline uint32_t primitive_stage_words(const DScene& S) {
    const unsigned long long w = (unsigned long long)S.n_planes * (sizeof(DPlane) / 4) + (unsigned long long)S.n_spheres * (sizeof(DSphere) / 4) +
                                 (unsigned long long)S.n_triangles * (sizeof(DTriangle) / 4);
    return w <= RT_STAGE_WORDS_MAX ? (uint32_t)w : 0u;
}
__host__ __device__ inline uint32_t cull_stage_words(const DScene& S) {
    return (primitive_stage_words(S) != 0 && S.n_range_nodes == 0 && S.n_instances <= 64) ? RT_CULL_WORDS * S.n_instances : 0u;
}
__host__ __device__ inline uint32_t analytic_stage_words(const DScene& S) { return primitive_stage_words(S) + cull_stage_words(S); }

}  // namespace

// ======================================================================================================
// Kernel 0: primary rays (raytracer.rs:30-51) for this shard's pixels -> ray list of depth 0.
// ======================================================================================================
// One launch generates the rays of the `n_here` local pixels from `first_pixel` on - the pixels of up to RT_VIEWS_PER_LAUNCH
// views, whose matrices V holds; a pass of more views takes several launches, and a block's list segment goes on where
// the launch before left it.
__global__ void __launch_bounds__(RT_BLOCK_THREADS) rt_generate(const DFrame F, const DViews V, const uint32_t first_pixel, const uint32_t n_here) {
    __shared__ uint32_t out_n;
    if (threadIdx.x == 0) out_n = F.counts[RT_ROW_RAYS * F.n_seg + blockIdx.x];       // (cleared when the frame starts)
    __syncthreads();
    const uint32_t lane = threadIdx.x & (RT_WAVE - 1);
    const uint32_t gwave = (blockIdx.x * blockDim.x + threadIdx.x) / RT_WAVE;
    const uint32_t n_waves = gridDim.x * (blockDim.x / RT_WAVE);
    for (uint32_t c = gwave; (unsigned long long)c * RT_WAVE < n_here; c += n_waves) {
        const uint32_t p = first_pixel + c * RT_WAVE + lane;
        uint32_t view = 0, col = 0, row = 0;
        const bool inside = c * RT_WAVE + lane < n_here && pixel_coords(F, p, view, col, row);
        RayRec rr;
        rr.ray.o = rr.ray.d = mk(0, 0, 0);
        // a wave's 64 pixels lie in one view (view_pixels is a multiple of 64): the matrix index is wave-uniform
        const uint32_t wave_view = __builtin_amdgcn_readfirstlane(c * RT_WAVE / F.view_pixels);      // among this launch's views
        if (inside) rr.ray = primary_ray(F, V.m[wave_view < RT_VIEWS_PER_LAUNCH ? wave_view : 0], col, row);
        rr.throughput = 1.0f;
        rr.meta = p;
        uint32_t slot = append_slot(inside, &out_n, F.seg_rays, F.n_seg, F.counts + RT_ROW_RAYS * F.n_seg, F.ovf_rays, F.overflow, F.overflow_seq, lane);
        if (slot != RT_REF_NONE) store_rec(F.rays[0], slot, rr);
    }
    __syncthreads();
    if (threadIdx.x == 0) F.counts[RT_ROW_RAYS * F.n_seg + blockIdx.x] = out_n < F.seg_rays ? out_n : F.seg_rays;
}

#ifdef RT_STATS
// Diagnostic build: per wave of the last rt_traverse launch: cycles, rounds, refill cycles, refills, cycles and
// count of rounds in which some lane tested a leaf, cycles and count of the other rounds, sum of walking lanes.
__device__ unsigned long long g_wave_timing[8192 * 10];
__device__ unsigned long long g_shade_timing[8192 * 8];
// per wave of the last launch that set rays up (rt_setup, or rt_shade with setup_next): cycles spent fetching the rays of its
// chunks, on the analytic primitives, on the culling boxes, pushing and draining pairs; chunks; drains
__device__ unsigned long long g_setup_timing[8192 * 8];
// shadow rays per depth: [2 * depth] set up, [2 * depth + 1] of them not traced (shadow_is_moot); never reset
__device__ unsigned long long g_moot_counts[2 * (RT_MAX_DEPTH_LIMIT + 2)];
extern "C" hipError_t rt_debug_moot_counts(unsigned long long* out, size_t n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_moot_counts), n * sizeof(unsigned long long));
}
extern "C" hipError_t rt_debug_setup_timing(unsigned long long* out, size_t n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_setup_timing), n * sizeof(unsigned long long));
}
extern "C" hipError_t rt_debug_shade_timing(unsigned long long* out, size_t n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_shade_timing), n * sizeof(unsigned long long));
}
__device__ unsigned int g_walk_items[64];
// walks of more than 300 node visits by one walker: {level, item, ray slot, instance, visits, 0, 0, 0, the item's 16 words}
#ifndef RT_DEBUG_LONG_LEVEL
#define RT_DEBUG_LONG_LEVEL 4
#endif
__device__ unsigned int g_long_walks[1 + 64 * 24];
extern "C" hipError_t rt_debug_long_walks(unsigned int* out, size_t n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_long_walks), n * sizeof(unsigned int));
}
extern "C" hipError_t rt_debug_walk_items(unsigned int* out, size_t n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_items), n * sizeof(unsigned int));
}
extern "C" hipError_t rt_debug_wave_timing(unsigned long long* out, size_t n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_timing), n * sizeof(unsigned long long));
}
#endif

// ======================================================================================================
// Kernel A: per ray, the analytic primitives (scene.rs:129-131) and the root test of every mesh instance.
// Writes the best analytic candidate to `best` and one walk item per instance whose root box the ray
// enters within the distance that still matters.  Rays with a non-finite component go to rt_setup_slow.
// The work runs either as a kernel of its own over all the rays of a depth (rt_setup) or, block by block,
// at the end of rt_shade over the rays that block has just made (see there).
// ======================================================================================================

// Copies the analytic primitives into LDS (every lane reads the same one: a broadcast, no scalar-load round trip per
// primitive).  Block-wide.
__device__ __forceinline__ void stage_analytic(const DScene& S, uint32_t* stage) {
    const uint32_t stage_words = analytic_stage_words(S);
    if (!stage_words) return;
    const uint32_t plane_words = S.n_planes * (sizeof(DPlane) / 4), sphere_words = S.n_spheres * (sizeof(DSphere) / 4);
    const uint32_t prim_words = primitive_stage_words(S);
    for (uint32_t w = threadIdx.x; w < stage_words; w += blockDim.x) {
        uint32_t v;
        if (w < plane_words) v = ((const uint32_t*)S.planes)[w];
        else if (w < plane_words + sphere_words) v = ((const uint32_t*)S.spheres)[w - plane_words];
        else if (w < prim_words) v = ((const uint32_t*)S.triangles)[w - plane_words - sphere_words];
        else {
            const uint32_t k = w - prim_words, f = k % RT_CULL_WORDS;
            const DInstance& inst = S.instances[k / RT_CULL_WORDS];
            v = f < 3 ? __float_as_uint(inst.world_lo[f]) : f < 6 ? __float_as_uint(inst.world_hi[f - 3]) :
                f == 6 ? __float_as_uint(inst.cull_origin_limit) : inst.root_ref;
        }
        stage[w] = v;
    }
    __syncthreads();
}

// LDS words behind the staged primitives: the rings of the block's waves (the scan of the range tree keeps no stack).
__device__ __host__ inline size_t setup_work_words(uint32_t /*range_depth*/, uint32_t threads) {
    return (size_t)(threads / RT_WAVE) * RT_RING_WORDS;
}

// Where a block finds the rays it sets up.
struct SetupInput {
    // rt_setup: all segments of the depth's lists, through a prefix table in LDS - one table, which holds the listed rays'
    // segments first and is built again for the hit records' when the block has done its chunks of listed rays (`table`,
    // with `table_scratch` for the scan)
    const uint32_t* pre; const uint32_t* pre_hits;
    uint32_t* table; uint32_t* table_scratch;
    uint32_t n_rays, n_hits;
    // rt_shade: what this block has just appended - the first own_* entries of its own segments and the slots it took from
    // the shared overflow segments (n_rays = own_rays + number of extra_rays, likewise for the hit records)
    uint32_t own_rays, own_hits;
    const uint32_t* extra_rays; const uint32_t* extra_hits;
};

// `stage`: the staged primitives (stage_analytic), followed by setup_work_words() of scratch.  `out_n`, `slow_n`: the block's
// LDS counters of walk items and of rays left to rt_setup_slow.  LOCAL: see SetupInput.
template <bool LOCAL>
__device__ __forceinline__ void setup_rays(const DScene& S, const DFrame& F, const uint32_t level, const SetupInput& in,
                                           uint32_t* stage, uint32_t* out_n_ptr, uint32_t* slow_n_ptr) {
    const LevelLists L = level_lists(F, level);
    const uint32_t p = level & 1;
    const uint32_t stage_words = analytic_stage_words(S);
    const bool staged = stage_words != 0;
    const uint32_t plane_words = S.n_planes * (sizeof(DPlane) / 4), sphere_words = S.n_spheres * (sizeof(DSphere) / 4);
    const DPlane* ld_planes = (const DPlane*)stage;
    const DSphere* ld_spheres = (const DSphere*)(stage + plane_words);
    const DTriangle* ld_triangles = (const DTriangle*)(stage + plane_words + sphere_words);
    const uint32_t cull_words = cull_stage_words(S);
    const uint32_t* ld_cull = stage + primitive_stage_words(S);
    const float4* rays = F.rays[p];
    const uint32_t lane = threadIdx.x & (RT_WAVE - 1);
    // the waves that share the input: all of the grid, or those of the block
    const uint32_t my_wave = LOCAL ? threadIdx.x / RT_WAVE : (blockIdx.x * blockDim.x + threadIdx.x) / RT_WAVE;
    const uint32_t n_waves = LOCAL ? blockDim.x / RT_WAVE : gridDim.x * (blockDim.x / RT_WAVE);
    PairRing R;
    R.ring = stage + stage_words + (threadIdx.x / RT_WAVE) * RT_RING_WORDS;
    R.head = 0; R.count = 0;
    uint32_t& out_n = *out_n_ptr;
    uint32_t& slow_n = *slow_n_ptr;
    uint32_t n_moot = 0;                       // shadow rays this lane did not trace (shadow_is_moot)
#ifdef RT_STATS
    uint32_t dbg_shadow_rays = 0;              // shadow rays this lane has looked at
#endif

    // Chunks of 64 rays: first the listed rays, then, light by light, the shadow rays of the hit records (a chunk
    // never mixes the two kinds, nor two lights).
    const uint32_t main_chunks = (in.n_rays + RT_WAVE - 1) / RT_WAVE, hit_chunks = (in.n_hits + RT_WAVE - 1) / RT_WAVE;
    const uint32_t all_chunks = main_chunks + S.n_lights * hit_chunks;
#ifdef RT_STATS
    unsigned long long su[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RT_SU_MARK(k) { const unsigned long long now_ = clock64(); su[k] += now_ - su_t; su_t = now_; }
#else
#define RT_SU_MARK(k)
#endif
    auto one_chunk = [&](const uint32_t c) {
#ifdef RT_STATS
        unsigned long long su_t = clock64();
        su[4] += 1;
#endif
        bool active;
        uint32_t gi = 0;
        RayRec rr;
        rr.ray.o = rr.ray.d = mk(0, 0, 0);
        rr.throughput = 0.0f;
        rr.meta = 0;
        if (c < main_chunks) {
            const uint32_t v = c * RT_WAVE + lane;
            active = v < in.n_rays;
            uint32_t at;
            if (LOCAL) at = v < in.own_rays ? blockIdx.x * F.seg_rays + v : (active ? in.extra_rays[v - in.own_rays] : 0u);
            else at = locate_chunk(in.pre, F.n_seg, F.seg_rays, c * RT_WAVE, in.n_rays, lane);
            if (active) gi = at;
        } else {
            const uint32_t cs = c - main_chunks, light = cs / hit_chunks, first = (cs - light * hit_chunks) * RT_WAVE;
            const uint32_t v = first + lane;
            active = v < in.n_hits;
            uint32_t hs;
            if (LOCAL) hs = v < in.own_hits ? blockIdx.x * F.seg_hits + v : (active ? in.extra_hits[v - in.own_hits] : 0u);
            else hs = locate_chunk(in.pre_hits, F.n_seg, F.seg_hits, first, in.n_hits, lane);
            if (active) {
                gi = F.shadow_base + hs * S.n_lights + light;
                const float4* hit = F.hits[p ^ 1] + 3 * (size_t)hs;
                const float4 h0 = hit[0], h1 = hit[1], h2 = hit[2];      // requested together: one trip to memory, not three
                rr = shadow_rec_of(S, h0, light);
                if (shadow_is_moot(S, F, h0, h1, h2, light, rr.ray.d)) {       // nobody needs this ray's verdict
                    L.best[gi] = RT_KEY_NONE;
                    active = false;
                    ++n_moot;
                }
            }
#ifdef RT_STATS
            dbg_shadow_rays += v < in.n_hits ? 1u : 0u;
#endif
        }
        if (active && c < main_chunks) rr = load_rec(rays, gi);
        const Ray r = rr.ray;
        const ShadowSpec ss = shadow_spec(S, rr);
        // Rays this kernel does not handle go to rt_setup_slow: non-finite components (the reference's exact order
        // matters for them), directions far from unit length or origins so far out that a local ray could
        // overflow (instance matrices of a cullable scene are below 1e15), or an instance without a culling box.
        const float o_l1 = fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z);
        const bool easy = finite3(r.o) && finite3(r.d) && fabsf(r.d.x) + fabsf(r.d.y) + fabsf(r.d.z) < 4.0f &&
                          S.all_cullable != 0 && o_l1 < 1e15f;
        {
            const uint32_t slot = append_slot(active && !easy, &slow_n, F.seg_rays, F.n_seg, L.slow_counts, F.ovf_rays, F.overflow, F.overflow_seq, lane);
            if (slot != RT_REF_NONE) F.slow_rays[slot] = gi;
            // whatever happens to such a ray (rt_setup_slow may not be launched for this frame, its list may be full - either
            // way the frame is invalid), the later kernels must find a well-formed key
            if (active && !easy) L.best[gi] = RT_KEY_NONE;
        }
        active = active && easy;
        const bool early = ss.is_shadow;

        // planes, spheres, triangles: intersect.rs:20-36 per kind, merged as scene.rs:129-131
        bool have = false; float best_t = 0.0f; uint32_t best_order = 0;
        bool settled = false;
        RT_SU_MARK(0)
        if (active) {
            if (staged) analytic_hits(S, ld_planes, ld_spheres, ld_triangles, r, early, ss, have, best_t, best_order, settled);
            else analytic_hits(S, S.planes, S.spheres, S.triangles, r, early, ss, have, best_t, best_order, settled);
        }
        RT_SU_MARK(1)
        // mesh instances in Vec order (scene.rs:133-136): a world-space culling box decides whether the ray can touch the
        // instance at all within the distance that still matters; the pairs that remain wait in the wave's ring for their
        // local ray and the reference's root-box test (ring_drain)
        const V3 inv_w = mk(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
        float lw = ss.limit_world;
        if (have && best_t < lw) lw = best_t;
        const float lw_cut = lw * 1.000244140625f;
        if (S.n_range_nodes) {
            // Many instances: the wave walks the range tree together.  The tree is stored in pre-order, so a ray's walk is a
            // forward scan that jumps over the subtrees whose box it misses; k is the first node any lane still has to look at
            // (lanes that are past it wait), and since every lane that is not at k has jumped over a subtree that contains k,
            // the next such node is k + 1 if a lane at k goes down into it and the end of k's subtree otherwise.  k is
            // wave-uniform: the nodes come through scalar loads, no lane keeps a stack, and every lane meets its instances in
            // Vec order.  (Fetching the two nodes the scan can go to next ahead of time made it slower.)
            uint32_t at = (active && !settled) ? 0u : S.n_range_nodes;     // the node this lane looks at next
            uint32_t k = __ballot(at == 0u) ? 0u : S.n_range_nodes;
            while (k < S.n_range_nodes) {
                const float4 n0 = S.range_nodes[2 * k], n1 = S.range_nodes[2 * k + 1];
                const uint32_t leaf_word = __float_as_uint(n0.w), end = __float_as_uint(n1.w);
                const bool here = at == k;
                float tmin_w;
                const bool pass = here && (!(o_l1 < S.range_limits[k]) ||
                                           (slab(n0.x, n0.y, n0.z, n1.x, n1.y, n1.z, r.o, inv_w, tmin_w) && !(tmin_w > lw_cut)));
                if (leaf_word & 0x80000000u) {
                    const bool want_item = pass && leaf_word != 0xFFFFFFFFu;
                    if (__ballot(want_item)) {
                        ring_push(R, want_item, r, lw, gi, leaf_word & 0x7FFFFFFFu, rr, ss, lane);
                        if (R.count >= RT_WAVE) ring_drain(R, S, F, L, &out_n, lane);
                    }
                    if (here) at = end;                                      // = k + 1
                    k = end;
                } else {
                    if (here) at = pass ? k + 1u : end;
                    k = __ballot(pass) ? k + 1u : end;
                }
            }
        }
        // Otherwise instance by instance.  With the culling records staged in LDS all boxes are tested first, in a loop of
        // nothing but broadcast reads and arithmetic (one bit per instance), and the pairs are pushed afterwards; as scalar
        // loads in the loop that also pushes, each instance cost a round trip to the scalar cache of its own - the four
        // boxes of the 1080p bunnies frame as much time as its seven analytic primitives.
        unsigned long long want_bits = 0;
        if (cull_words && active && !settled) {
#pragma unroll 4
            for (uint32_t ii = 0; ii < S.n_instances; ++ii) {
                const uint32_t* c = ld_cull + RT_CULL_WORDS * ii;
                float tmin_w;
                const bool w = c[7] != RT_REF_NONE &&
                               (!(o_l1 < __uint_as_float(c[6])) ||
                                (slab(__uint_as_float(c[0]), __uint_as_float(c[1]), __uint_as_float(c[2]), __uint_as_float(c[3]),
                                      __uint_as_float(c[4]), __uint_as_float(c[5]), r.o, inv_w, tmin_w) && !(tmin_w > lw_cut)));
                want_bits |= (w ? 1ull : 0ull) << ii;
            }
        }
        RT_SU_MARK(2)
        for (uint32_t ii = 0; ii < (S.n_range_nodes ? 0u : S.n_instances); ++ii) {
            if (cull_words) {
                const bool want_item = (want_bits >> ii) & 1ull;
                if (!__ballot(want_item)) continue;
                ring_push(R, want_item, r, lw, gi, ii, rr, ss, lane);
                if (R.count >= RT_WAVE) ring_drain(R, S, F, L, &out_n, lane);
                continue;
            }
            const DInstance& inst = S.instances[ii];
            bool want_item = false;
            if (active && !settled && inst.root_ref != RT_REF_NONE) {
                // the box is only trusted for origins within cull_origin_limit; farther rays always get the exact test
                float tmin_w;
                want_item = !(o_l1 < inst.cull_origin_limit) ||
                            (slab(inst.world_lo[0], inst.world_lo[1], inst.world_lo[2], inst.world_hi[0], inst.world_hi[1],
                                  inst.world_hi[2], r.o, inv_w, tmin_w) && !(tmin_w > lw_cut));
            }
            if (!__ballot(want_item)) continue;
            ring_push(R, want_item, r, lw, gi, ii, rr, ss, lane);
            if (R.count >= RT_WAVE) ring_drain(R, S, F, L, &out_n, lane);
        }
        if (active) L.best[gi] = have ? make_key(best_t, best_order) : RT_KEY_NONE;
        RT_SU_MARK(3)
    };
    // Chunk c goes to wave c mod n_waves.  rt_setup does it in two phases, listed rays and hit records, with the table built
    // again in between (two loops rather than one loop nest: as a nest the kernel spilled 23 registers).
    if (LOCAL) {
        for (uint32_t c = my_wave; c < all_chunks; c += n_waves) one_chunk(c);
    } else {
        for (uint32_t c = my_wave; c < main_chunks; c += n_waves) one_chunk(c);
        __syncthreads();                                               // every wave of the block is done with the rays' table
        build_prefix(L.hit_counts, F.n_seg, F.seg_hits, F.ovf_hits, in.table, in.table_scratch);
        for (uint32_t c = main_chunks + (my_wave + n_waves - main_chunks % n_waves) % n_waves; c < all_chunks; c += n_waves) one_chunk(c);
    }
#ifdef RT_STATS
    {
        const uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) / RT_WAVE;
        if (lane == 0 && gw < 8192) for (int k = 0; k < 8; ++k) g_setup_timing[gw * 8 + k] = su[k];
    }
#endif
    while (R.count) ring_drain(R, S, F, L, &out_n, lane);           // the pairs still waiting
    {   // column 0 of the wave's row of statistics (the grid's waves own a row each, see rt_shade)
        for (int off = 32; off > 0; off >>= 1) n_moot += __shfl_down(n_moot, off);
#ifdef RT_STATS
        for (int off = 32; off > 0; off >>= 1) dbg_shadow_rays += __shfl_down(dbg_shadow_rays, off);
        if (lane == 0 && dbg_shadow_rays) {                        // one pair of atomics per wave
            atomicAdd(&g_moot_counts[2 * level], (unsigned long long)dbg_shadow_rays);
            atomicAdd(&g_moot_counts[2 * level + 1], (unsigned long long)n_moot);
        }
#endif
        if (lane == 0 && n_moot) F.wave_stats[(size_t)((blockIdx.x * blockDim.x + threadIdx.x) / RT_WAVE) * 8] += n_moot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        L.fat_counts[blockIdx.x] = out_n < F.seg_fat ? out_n : F.seg_fat;
        L.slow_counts[blockIdx.x] = slow_n < F.seg_rays ? slow_n : F.seg_rays;
        if (slow_n) {
            F.slow_flags[level] = 1u;
            if (!F.slow_launched) *F.needs_slow = F.overflow_seq;      // nobody will trace these rays: the frame has to be redone
        }
    }
}

// five waves per SIMD (the list kernels run five blocks per compute unit): left to itself the compiler takes 113 registers,
// four waves' worth, and the kernel, which mostly waits for memory, runs a third longer
__attribute__((amdgpu_waves_per_eu(5)))
__global__ void __launch_bounds__(RT_SETUP_THREADS) rt_setup(const DScene S, const DFrame F, const uint32_t level) {
    const LevelLists L = level_lists(F, level);
    extern __shared__ uint32_t lds[];
    uint32_t* pre = lds;                                   // n_seg + 1: the listed rays' segments, later the hit records' (setup_rays)
    uint32_t* scratch = lds + F.n_seg + 1;                 // blockDim.x
    uint32_t* stage = scratch + blockDim.x;
    __shared__ uint32_t out_n, slow_n;
    if (threadIdx.x == 0) { out_n = 0; slow_n = 0; }
    SetupInput in;
    in.pre = pre; in.pre_hits = pre;
    in.table = pre; in.table_scratch = scratch;
    // (the hit records of the depth above, whose shadow rays are traced at this depth: only their number for now)
    in.n_hits = build_prefix(L.hit_counts, F.n_seg, F.seg_hits, F.ovf_hits, pre, scratch);
    in.n_rays = build_prefix(L.ray_counts, F.n_seg, F.seg_rays, F.ovf_rays, pre, scratch);
    in.own_rays = in.own_hits = 0; in.extra_rays = in.extra_hits = nullptr;
    stage_analytic(S, stage);
    setup_rays<false>(S, F, level, in, stage, &out_n, &slow_n);
}

// ======================================================================================================
// Kernel A': the rays rt_setup handed over (none, for ordinary scenes).  Everything Scene::intersect does for
// such a ray except the BVH walks of finite local rays: analytic primitives, per instance the local ray
// (intersect.rs:296-308) and either the whole walk in the reference's own order (a ray with a non-finite
// component) or the exact root test -> BVH work item.
// ======================================================================================================
__global__ void __launch_bounds__(RT_BLOCK_THREADS) rt_setup_slow(const DScene S, const DFrame F, const uint32_t level) {
    const LevelLists L = level_lists(F, level);
    if (F.slow_flags[level] == 0) return;                          // the usual case: rt_setup kept every ray
    extern __shared__ uint32_t lds[];
    uint32_t* pre = lds;                       // n_seg + 1
    uint32_t* scratch = lds + F.n_seg + 1;     // blockDim.x
    __shared__ uint32_t out_n;
    if (threadIdx.x == 0) out_n = L.fat_counts[blockIdx.x];         // this block's segment continues where rt_setup stopped
    const uint32_t n_rays = build_prefix(L.slow_counts, F.n_seg, F.seg_rays, F.ovf_rays, pre, scratch);
    // the analytic primitives are staged in LDS (every lane reads the same one: a broadcast, no scalar-load
    // round trip per primitive)
    uint32_t* stage = lds + F.n_seg + 1 + blockDim.x;
    const uint32_t stage_words = analytic_stage_words(S);
    const bool staged = stage_words != 0;
    const uint32_t plane_words = S.n_planes * (sizeof(DPlane) / 4), sphere_words = S.n_spheres * (sizeof(DSphere) / 4);
    const DPlane* ld_planes = (const DPlane*)stage;
    const DSphere* ld_spheres = (const DSphere*)(stage + plane_words);
    const DTriangle* ld_triangles = (const DTriangle*)(stage + plane_words + sphere_words);
    stage_analytic(S, stage);
    const uint32_t lane = threadIdx.x & (RT_WAVE - 1);
    const uint32_t gwave = (blockIdx.x * blockDim.x + threadIdx.x) / RT_WAVE;
    const uint32_t n_waves = gridDim.x * (blockDim.x / RT_WAVE);

    for (uint32_t c = gwave; (unsigned long long)c * RT_WAVE < n_rays; c += n_waves) {
        const uint32_t v = c * RT_WAVE + lane;
        const bool active = v < n_rays;
        uint32_t gi = 0;
        RayRec rr;
        rr.ray.o = rr.ray.d = mk(0, 0, 0);
        rr.throughput = 0.0f;
        rr.meta = 0;
        if (active) { gi = F.slow_rays[locate(pre, F.n_seg, F.seg_rays, v)]; rr = load_any_rec(S, F, level, gi); }
        const Ray r = rr.ray;
        const ShadowSpec ss = shadow_spec(S, rr);
        const bool ray_finite = finite3(r.o) && finite3(r.d);
        const bool early = ss.is_shadow && (ss.directional || ray_finite);

        // planes, spheres, triangles: intersect.rs:20-36 per kind, merged as scene.rs:129-131
        bool have = false; float best_t = 0.0f; uint32_t best_order = 0;
        bool settled = false;
        if (active) {
            if (staged) analytic_hits(S, ld_planes, ld_spheres, ld_triangles, r, early, ss, have, best_t, best_order, settled);
            else analytic_hits(S, S.planes, S.spheres, S.triangles, r, early, ss, have, best_t, best_order, settled);
        }

        // mesh instances in Vec order (scene.rs:133-136)
        bool exact_mesh = false;             // exact path: the winner so far is a mesh triangle
        float ex_t = 0.0f, ex_u = 0.0f, ex_v = 0.0f; uint32_t ex_slot = 0, ex_inst = 0;
        // A candidate with t = NaN (a local ray with a NaN direction, as a singular instance matrix makes it) is only
        // taken when nothing was found before it and then blocks everything after it (utils/mod.rs:17): what it does
        // depends on the outcome of the instances before it.  So a ray with one such instance walks all of them here,
        // one after the other in Vec order; only rays whose local rays are all finite leave their walks to rt_traverse.
        bool all_here = !ray_finite;
        if (active && ray_finite && !settled)
            for (uint32_t ii = 0; ii < S.n_instances; ++ii) {
                const DInstance& inst = S.instances[ii];
                if (inst.root_ref == RT_REF_NONE) continue;
                const V3 lo = transform34(inst.model_inverse, r.o);
                const V3 dv = transform34(inst.rotation_scale_inverse, r.d);
                const float dn = sqrtf(dot3(dv, dv));
                if (!(finite3(lo) && finite3(mk(dv.x / dn, dv.y / dn, dv.z / dn)))) all_here = true;
            }
        for (uint32_t ii = 0; ii < S.n_instances; ++ii) {
            const DInstance& inst = S.instances[ii];
            bool want_item = false;
            if (active && !settled && inst.root_ref != RT_REF_NONE && !(have && best_t != best_t)) {
                Ray lr;
                lr.o = transform34(inst.model_inverse, r.o);
                V3 dv = transform34(inst.rotation_scale_inverse, r.d);
                float dn = sqrtf(dot3(dv, dv));
                lr.d = mk(dv.x / dn, dv.y / dn, dv.z / dn);                 // glm::normalize, intersect.rs:304-308
                if (all_here) {
                    // reference order, to the end, right here
                    bool m_have = false; float m_t = 0.0f, m_u = 0.0f, m_v = 0.0f; uint32_t m_slot = 0;
                    // The reference tests every node's own box, the root's included (intersect.rs:247-250).  The walk below
                    // tests a node's children's boxes when it visits the node, so the root's box has to be tested here: a
                    // mesh of a single leaf has no other box, and with NaN in the ray - or in a tree whose boxes do not
                    // nest - a child's box can pass where the root's does not.
                    float t_root;
                    if (slab(inst.root_lo[0], inst.root_lo[1], inst.root_lo[2], inst.root_hi[0], inst.root_hi[1], inst.root_hi[2], lr.o,
                             mk(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z), t_root))
                        exact_mesh_hit(S, inst.root_ref, lr, m_have, m_t, m_u, m_v, m_slot);
                    if (m_have) {
                        float tw = world_t(S, inst, lr, r.o, m_t, m_slot);
                        if (closer(have, best_t, tw)) {
                            have = true; best_t = tw; exact_mesh = true;
                            ex_t = m_t; ex_u = m_u; ex_v = m_v; ex_slot = m_slot; ex_inst = ii;
                        }
                        settled = early && have && (ss.directional || best_t < ss.limit_world);
                    }
                } else {
                    want_item = true;
                }
            }
            // beyond the best analytic hit (or the light) nothing in this mesh can matter; see rt_traverse
            float lw = ss.limit_world;
            if (have && best_t < lw) lw = best_t;
            float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
            if (want_item) want_item = walk_item(inst, ii, gi, r, ss, lw, q0, q1, q2, q3);
            const uint32_t slot = append_slot(want_item, &out_n, F.seg_fat, F.n_seg, L.fat_counts, F.ovf_fat, F.overflow, F.overflow_seq, lane);
            if (slot != RT_REF_NONE) store_walk_item(L, slot, q0, q1, q2, q3);
        }
        // a mesh winner found on the exact path is recorded as a finished work item
        {
            uint32_t slot = append_slot(active && exact_mesh, &out_n, F.seg_fat, F.n_seg, L.fat_counts, F.ovf_fat, F.overflow, F.overflow_seq, lane);
            if (slot != RT_REF_NONE) {
                const float4 z = make_float4(0, 0, 0, 0);
                store_walk_item(L, slot, z, z, z, make_float4(__uint_as_float(gi), __uint_as_float(ex_inst | RT_FAT_DONE), 0.0f, 0.0f));
                L.item_results[slot] = make_float4(ex_t, ex_u, ex_v, __uint_as_float(ex_slot));
                best_order = RT_KEY_MESH | slot;
            } else if (active && exact_mesh) {
                have = false;                                              // list overflow: the frame is redone anyway
            }
        }
        if (active) {
            // finite rays: ordered key, merged with the work items' results by kernel B; rays traced on the exact
            // path are final here and keep the raw bits of t (which may be NaN)
            unsigned long long key = RT_KEY_NONE;
            if (have) key = (ray_finite && !exact_mesh) ? make_key(best_t, best_order)
                                                        : (((unsigned long long)__float_as_uint(best_t) << 32) | best_order);
            L.best[gi] = key;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) L.fat_counts[blockIdx.x] = out_n < F.seg_fat ? out_n : F.seg_fat;
}

// ======================================================================================================
// First part of kernel B, done by its first blocks: the pairs walk_item left on the depth's heavy list - rays that enter a
// mesh's root box so far out that the slab test (intersect.rs:185-226) no longer tells the mesh's boxes apart: nearly
// every box "passes", the reference tests nearly every triangle (82 000 of them for one shadow ray of the 1080p bunnies
// frame, seen from a hit point 5e8 units down the mirror floor), and a lane-by-lane walk of such a pair keeps one wave
// busy for milliseconds while the launch waits.  The list is complete before the launch starts (rt_setup or rt_shade
// wrote it), so the blocks that take it need nobody else's work; the other blocks meanwhile start on the walk items,
// which are handed out chunk by chunk.  A block takes one pair at a time and its lanes take the mesh's leaves side by
// side: a leaf counts when the ray passes the leaf's own box - in a tree whose boxes nest that is the case exactly when
// it passes every box above it as well - and its triangles are tested as in the walk; the closest candidate by (t, slot)
// is what any visiting order yields.
// ======================================================================================================
#define RT_HEAVY_BLOCKS 32
// Block-wide: every thread of the first `heavy_blocks` blocks of rt_traverse calls it.
__device__ void traverse_heavy_pairs(const DScene& S, const DFrame& F, const uint32_t level, const uint32_t heavy_blocks) {
    const LevelLists L = level_lists(F, level);
    const uint32_t n = *L.heavy_count < RT_HEAVY_CAP ? *L.heavy_count : RT_HEAVY_CAP;
    __shared__ unsigned long long best_key;
    uint32_t n_boxes = 0, n_triangle_tests = 0;
    for (uint32_t e = blockIdx.x; e < n; e += heavy_blocks) {
        if (threadIdx.x == 0) best_key = RT_KEY_NONE;
        __syncthreads();
        const uint32_t item = L.heavy_items[e];
        const float4 q0 = L.fat_items[4 * (size_t)item], q1 = L.fat_items[4 * (size_t)item + 1], q3 = L.fat_items[4 * (size_t)item + 3];
        Ray lr;
        lr.o = mk(q0.x, q0.y, q0.z);
        lr.d = mk(q1.x, q1.y, q1.z);
        const V3 inv = mk(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
        const uint32_t gi = __float_as_uint(q3.x), ii = __float_as_uint(q3.y) & RT_FAT_INSTANCE;
        const DInstance& inst = S.instances[ii];
        bool m_have = false; float m_t = 0.0f; uint32_t m_slot = 0;
        // (the next leaf's box is requested before this leaf's triangles, and a leaf's triangles all together: two round
        // trips to memory per leaf instead of six)
        const float4* boxes = S.leaf_boxes + 2 * (size_t)inst.leaf_begin;
        uint32_t k = threadIdx.x;
        float4 b0 = make_float4(0, 0, 0, 0), b1 = b0;
        if (k < inst.leaf_count) { b0 = boxes[2 * (size_t)k]; b1 = boxes[2 * (size_t)k + 1]; }
        while (k < inst.leaf_count) {
            const float4 c0 = b0, c1 = b1;
            k += blockDim.x;
            if (k < inst.leaf_count) { b0 = boxes[2 * (size_t)k]; b1 = boxes[2 * (size_t)k + 1]; }
            float tmin;
            ++n_boxes;
            if (!slab(c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, lr.o, inv, tmin)) continue;
            const uint32_t ref = __float_as_uint(c1.z), first = ref & RT_REF_SLOT_MASK, cnt = (ref >> RT_REF_COUNT_SHIFT) & 7u;
            n_triangle_tests += cnt;
            float4 h[RT_LEAF_TRIANGLES][3];
#pragma unroll
            for (uint32_t j = 0; j < RT_LEAF_TRIANGLES; ++j) {
                const uint32_t slot = first + (j < cnt ? j : cnt - 1);          // (entries past the leaf's count repeat its last triangle)
                h[j][0] = S.tri_hot[3 * slot]; h[j][1] = S.tri_hot[3 * slot + 1]; h[j][2] = S.tri_hot[3 * slot + 2];
            }
#pragma unroll
            for (uint32_t j = 0; j < RT_LEAF_TRIANGLES; ++j) {
                const uint32_t slot = first + j;
                float t, u, v;
                if (j < cnt && triangle_tuv(mk(h[j][0].x, h[j][0].y, h[j][0].z), mk(h[j][0].w, h[j][1].x, h[j][1].y), mk(h[j][1].z, h[j][1].w, h[j][2].x), lr, t, u, v) &&
                    (!m_have || t < m_t || (t == m_t && slot < m_slot))) {
                    m_have = true; m_t = t; m_slot = slot;
                }
            }
        }
        if (m_have) atomicMin(&best_key, make_key(m_t, m_slot));
        __syncthreads();
        if (threadIdx.x == 0 && best_key != RT_KEY_NONE) {
            // the pair's closest hit joins the ray's other candidates as in rt_traverse (scene.rs:133-136)
            const uint32_t win_slot = (uint32_t)best_key;
            const float4 h0 = S.tri_hot[3 * win_slot], h1 = S.tri_hot[3 * win_slot + 1], h2 = S.tri_hot[3 * win_slot + 2];
            float t = 0.0f, u = 0.0f, v = 0.0f;
            triangle_tuv(mk(h0.x, h0.y, h0.z), mk(h0.w, h1.x, h1.y), mk(h1.z, h1.w, h2.x), lr, t, u, v);
            const float tw = world_t(S, inst, lr, load_any_origin(S, F, level, gi), t, win_slot);
            L.item_results[item] = make_float4(t, u, v, __uint_as_float(win_slot));
            atomicMin(&L.best[gi], make_key(tw, RT_KEY_MESH | item));
        }
        __syncthreads();
    }
    // work statistics, in the rows of the walk's counters (four boxes count as one node visit)
    if (n) {
        const uint32_t lane = threadIdx.x & (RT_WAVE - 1), gwave = (blockIdx.x * blockDim.x + threadIdx.x) / RT_WAVE;
        uint32_t a = n_boxes, b = n_triangle_tests;
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
        if (lane == 0 && (a | b)) { F.wave_stats[(size_t)gwave * 8 + 5] += (a + 3) / 4; F.wave_stats[(size_t)gwave * 8 + 6] += b; }
        if (threadIdx.x == 0 && blockIdx.x < n) F.wave_stats[(size_t)gwave * 8 + 7] += (n - blockIdx.x + heavy_blocks - 1) / heavy_blocks;      // the pairs this block took
    }
}


// ======================================================================================================
// Kernel B: BVH traversal of the work items, one (ray, instance) pair per lane, lanes refilled as they
// finish.  The outcome is that of intersect.rs:228-326 for the pair; how it gets there:
//  * A node's box is tested when its parent is visited (the parent holds both child boxes), with the
//    reference's arithmetic, so the set of boxes that "pass" is the reference's.
//  * Distance pruning: a box whose entry distance lies beyond the best candidate so far (for this pair,
//    for the ray as a whole, or beyond the light of a shadow ray) by more than the slack cannot hold a
//    candidate that compares lower.  The slack (RT_PRUNE_REL and the instance's prune_abs) dominates
//    the rounding error of Moeller-Trumbore's t for the triangle sizes of a `prunable` mesh; other
//    meshes are walked exactly as the reference walks them.
//  * Near child first.  The reference resolves equal t by visiting order (utils/mod.rs:17, strict `<`);
//    triangle slots are numbered in that visiting order, so "lower slot wins a tie" gives the same
//    winner in any visiting order.
//  * Shadow rays (raytracer.rs:171-187) only need `t < distance to the light`: the walk stops at the
//    first candidate that settles it.
// The per-lane stack lives in LDS: entry e of lane l is stack[e * 64 + l] (child reference) and
// stack[(entries + e) * 64 + l] (box entry distance).
// ======================================================================================================

__global__ void __launch_bounds__(RT_BLOCK_THREADS) rt_traverse(const DScene S, const DFrame F, const uint32_t level) {
#ifdef RT_STATS
    const unsigned long long t_begin = clock64(), w_begin = wall_clock64();
#endif
    const LevelLists L = level_lists(F, level);
    {
        const uint32_t heavy_blocks = gridDim.x < RT_HEAVY_BLOCKS ? gridDim.x : RT_HEAVY_BLOCKS;
        if (blockIdx.x < heavy_blocks && *L.heavy_count != 0) traverse_heavy_pairs(S, F, level, heavy_blocks);
    }
    extern __shared__ uint32_t lds[];
    uint32_t* pre = lds;
    uint32_t* scratch = lds + F.n_seg + 1;
    const uint32_t n_items = build_prefix(L.fat_counts, F.n_seg, F.seg_fat, F.ovf_fat, pre, scratch);
    const uint32_t lane = threadIdx.x & (RT_WAVE - 1);
    const uint32_t wave = threadIdx.x / RT_WAVE;
    uint32_t* wave_stack = lds + F.n_seg + 1 + blockDim.x + wave * (2 * S.stack_lds * RT_WAVE);
    uint32_t* stack = wave_stack + lane;
    uint32_t* stack_t = stack + S.stack_lds * RT_WAVE;
    // entries beyond the LDS part (a walk that keeps many siblings pending at once: rare) live in device memory
    uint2* spill = F.spill + (blockIdx.x * blockDim.x + threadIdx.x);
    const uint32_t stack_lds = S.stack_lds;
    const size_t spill_lanes = F.spill_lanes;
    // one record per lane for the pair that lane started: {key lo = triangle slot, key hi = bits of local t,
    // walkers still at work on the pair, "shadow proven"}; 16-byte aligned
    uint32_t* recs = lds + ((F.n_seg + 1 + blockDim.x + (blockDim.x / RT_WAVE) * (2 * S.stack_lds * RT_WAVE) + 3u) & ~3u) +
                     wave * (4 * RT_WAVE);
    uint32_t* donor_list = scratch + wave * RT_WAVE;               // free again once the prefix table is built
    const uint32_t gwave = (blockIdx.x * blockDim.x + threadIdx.x) / RT_WAVE;
    const uint32_t n_waves = gridDim.x * (blockDim.x / RT_WAVE);

    // Input is handed out in chunks of 64 pairs: every wave starts with the chunk of its own index, further ones
    // come from a counter in device memory (one atomic per 64 pairs), so that waves with light pairs take more.
    // With less than 64 pairs per wave the chunks shrink, so that every wave gets some and shares the long ones.
    const bool one_chunk_each = (unsigned long long)n_waves * RT_WAVE >= n_items;
    const uint32_t n_chunks = one_chunk_each ? n_waves : (n_items + RT_WAVE - 1) / RT_WAVE;
    // neighbouring pairs walk the same parts of the same meshes.  Interleaved over the waves they even out what a wave has to
    // do, which is what a launch with a chunk or two per wave needs; handed out in runs they keep a wave's fetches in few
    // cache lines, which pays once a wave gets several chunks of a launch (F.walk_contiguous == 2), or always in scenes far
    // larger than the caches (1)
    const bool contiguous = F.walk_contiguous == 1 ||
                            (F.walk_contiguous == 2 && n_items >= (unsigned long long)RT_CONTIGUOUS_MIN_CHUNKS * n_waves * RT_WAVE);
    const uint32_t chunk_len = (n_items + n_chunks - 1) / n_chunks;
    uint32_t my_chunk = gwave, cur_chunk = 0, chunk_next = 0, chunk_end = 0;       // wave-uniform
    bool exhausted = false;

    // per-lane state of the walk
    bool running = false;           // walking a subtree of a pair
    bool pending = false;           // walk finished, its part in the pair's result not yet settled
    uint32_t item = 0, gi = 0, ii = 0, sp = 0, base = 0, owner = lane;
    uint32_t pending_leaf = RT_REF_NONE;                            // a leaf this lane has reached and not yet tested
    // The entry the lane looks at next is already off the stack: its reference and its box entry distance.  The node
    // itself is fetched when it is visited: requesting it a step ahead kept 28 more registers alive through the leaf tests
    // and the service phase, and the kernel at three waves per SIMD; with the leaves tested in two groups as well it needs
    // 112 registers and runs four (BVH walk of the 1080p bunnies frame 1.22 -> 1.16 ms, of the 2.6 M-triangle grid 6.25 -> 5.73).
    uint32_t cur_ref = RT_REF_NONE;
    float cur_t = 0.0f;
    Ray lr; lr.o = lr.d = mk(0, 0, 0);
    V3 inv = mk(0, 0, 0);
    bool fast = false;
    float limit = RT_INF, prune_abs = 0.0f, shadow_thr = -RT_INF, inv_dn = 0.0f;
    bool m_have = false; float m_t = 0.0f, m_u = 0.0f, m_v = 0.0f; uint32_t m_slot = 0;
    uint32_t n_node_visits = 0, n_triangle_tests = 0;      // this lane's share of the launch's work, for the statistics
#ifdef RT_STATS
    uint32_t dbg_steps = 0, dbg_walk_steps = 0;
    unsigned long long tm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    tm[2] = clock64() - t_begin;           // building the prefix table
    if (threadIdx.x == 0 && blockIdx.x == 0 && level < 64) g_walk_items[level] = n_items;
    tm[3] = w_begin;                       // start and end of the wave on the constant 100 MHz clock
#endif

    // A depth-first walk of any mesh of the scene fits S.stack_entries entries (computed from the trees when the scene was
    // created).  Should a walk ever ask for more - a corrupted list entry, a tree that is not what was uploaded - the entry
    // is dropped and the frame flagged like one whose lists overflowed, instead of writing past the stack: the reference's
    // own walk fails loudly in that place too (ArrayVec::push panics, intersect.rs:240-255).
    const uint32_t stack_entries = S.stack_entries;
    auto stack_push = [&](uint32_t e, uint32_t ref, float t) {
        if (e < stack_lds) { stack[e * RT_WAVE] = ref; stack_t[e * RT_WAVE] = __float_as_uint(t); }
        else if (e < stack_entries) spill[(e - stack_lds) * spill_lanes] = make_uint2(ref, __float_as_uint(t));
        else *F.overflow = F.overflow_seq;
    };

    for (;;) {
        const unsigned long long idle_mask = __ballot(!running);
        const unsigned long long run_mask = ~idle_mask;
        const bool out_of_input = exhausted && chunk_next == chunk_end;
#ifdef RT_STATS
        const unsigned long long t_round = clock64();
        tm[1] += 1;
        tm[8] += __popcll(run_mask);
        bool serviced = false;
#endif
        // ---- sharing, part 1: once this wave has no more input, idle lanes take over the oldest pending subtree
        // of lanes that are still walking, so that one long walk does not keep 63 lanes (and the kernel) waiting
        bool can_give = false;
        unsigned long long give_mask = 0;
        if (out_of_input && __popcll(idle_mask) >= RT_SHARE_IDLE && run_mask != 0) {
            can_give = running && fast && sp - base >= 1 && (sp - base >= 2 || cur_ref != RT_REF_NONE || pending_leaf != RT_REF_NONE) &&
                       base < stack_lds && !(stack[base * RT_WAVE] & RT_REF_LEAF_BIT);
            give_mask = __ballot(can_give);
        }
        // ---- service: once a quarter of the wave is idle, nobody is walking, or there is something to share ---
        if (run_mask == 0 || (!out_of_input && __popcll(idle_mask) >= RT_REFILL_IDLE) || give_mask != 0) {
#ifdef RT_STATS
            serviced = true;
#endif
            // (1) finished walkers leave their pair; the last one to leave merges the pair's closest hit into the
            // ray's best candidate (scene.rs:133-136).  Every walker has merged its own closest hit into the
            // pair's record by (t, slot), the order a single walker applies.
            if (pending) {
                pending = false;
                if (atomicSub(&recs[4 * owner + 2], 1u) == 1u) {
                    const uint32_t win_slot = __atomic_load_n(&recs[4 * owner], __ATOMIC_RELAXED);
                    if (__atomic_load_n(&recs[4 * owner + 1], __ATOMIC_RELAXED) != 0xFFFFFFFFu) {
                        if (!(m_have && m_slot == win_slot)) {
                            // another walker found it: the test is a pure function of ray and triangle
                            const float4 h0 = S.tri_hot[3 * win_slot], h1 = S.tri_hot[3 * win_slot + 1], h2 = S.tri_hot[3 * win_slot + 2];
                            triangle_tuv(mk(h0.x, h0.y, h0.z), mk(h0.w, h1.x, h1.y), mk(h1.z, h1.w, h2.x), lr, m_t, m_u, m_v);
                            m_slot = win_slot;
                        }
                        float tw = world_t(S, S.instances[ii], lr, load_any_origin(S, F, level, gi), m_t, m_slot);      // from the ray's world origin
                        L.item_results[item] = make_float4(m_t, m_u, m_v, __uint_as_float(m_slot));
                        atomicMin(&L.best[gi], make_key(tw, RT_KEY_MESH | item));
                    }
                }
            }
            // (2) new pairs from the input
            if (chunk_next == chunk_end && !exhausted) {
                if (my_chunk == 0xFFFFFFFFu) {
                    uint32_t taken = 0;
                    if (!one_chunk_each) {
                        if (lane == 0) taken = atomicAdd(&F.chunk_counter[level], 1u);
                        taken = __builtin_amdgcn_readfirstlane(taken);
                    }
                    my_chunk = n_waves + taken;
                }
                // chunk c holds the pairs c, c + n_chunks, c + 2 * n_chunks, ... (or, for scenes far larger than the caches,
                // the run c * chunk_len .. of neighbouring pairs, which walk the same meshes)
                if (my_chunk >= n_chunks || (contiguous ? (unsigned long long)my_chunk * chunk_len : my_chunk) >= n_items) exhausted = true;
                else {
                    cur_chunk = my_chunk;
                    chunk_next = 0;
                    chunk_end = contiguous ? (n_items - my_chunk * chunk_len < chunk_len ? n_items - my_chunk * chunk_len : chunk_len)
                                           : (n_items - my_chunk + n_chunks - 1) / n_chunks;
                    my_chunk = 0xFFFFFFFFu;                          // the next one is asked for when it is needed
                }
            }
            const uint32_t avail = chunk_end - chunk_next;
            if (avail == 0 && run_mask == 0) break;                 // exhausted and nothing in flight
            const uint32_t asking = __popcll(idle_mask);
            const uint32_t rank = __popcll(idle_mask & ((1ull << lane) - 1ull));
            const bool takes = !running && rank < avail;
            // a run of neighbouring pairs lies in one list segment or a few: the wave finds the first one's together
            uint32_t run_slot = 0;
            if (contiguous && avail != 0 && asking != 0)
                run_slot = locate_run(pre, F.n_seg, F.seg_fat, cur_chunk * chunk_len + chunk_next, rank, takes, lane);
            if (!running) {
                if (rank < avail) {
                    // Element j of chunk c is pair c + j * n_chunks: neighbouring pairs (similar rays, similar cost) go to
                    // different chunks.  A 64-byte item is a cache line of its own either way.
                    const size_t fat = contiguous ? run_slot : locate(pre, F.n_seg, F.seg_fat, cur_chunk + (chunk_next + rank) * n_chunks);
                    const float4 q0 = L.fat_items[4 * fat], q1 = L.fat_items[4 * fat + 1], q2 = L.fat_items[4 * fat + 2],
                                 q3 = L.fat_items[4 * fat + 3];
                    // (an entry of a ray traced on the exact path only records a result, see rt_setup_slow: there is nothing to walk)
                    // (nor in an entry left to traverse_heavy_pairs)
                    if (!(__float_as_uint(q3.y) & (RT_FAT_DONE | RT_FAT_HEAVY))) {
                    lr.o = mk(q0.x, q0.y, q0.z);
                    lr.d = mk(q1.x, q1.y, q1.z);
                    inv = mk(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
                    limit = q1.w;
                    inv_dn = q2.x; shadow_thr = q2.y; prune_abs = q2.z;
                    gi = __float_as_uint(q3.x);
                    ii = __float_as_uint(q3.y) & RT_FAT_INSTANCE;
                    fast = (__float_as_uint(q3.y) & RT_FAT_FAST) != 0;
                    item = (uint32_t)fat;
#ifdef RT_STATS
                    dbg_walk_steps = 0;
#endif
                    m_have = false; m_t = 0.0f; m_u = 0.0f; m_v = 0.0f; m_slot = 0;
                    cur_ref = __float_as_uint(q2.w);
                    cur_t = q0.w;
                    sp = 0; base = 0;
                    owner = lane;
                    recs[4 * lane] = 0xFFFFFFFFu; recs[4 * lane + 1] = 0xFFFFFFFFu;
                    recs[4 * lane + 2] = 1u; recs[4 * lane + 3] = 0u;
                    running = true;
                    }
                }
            }
            chunk_next += asking < avail ? asking : avail;
            // (3) sharing, part 2 (no input was left, so the idle lanes are still idle)
            if (give_mask) {
                const uint32_t n_idle = __popcll(idle_mask), n_give = __popcll(give_mask);
                const uint32_t n = n_idle < n_give ? n_idle : n_give;
                const uint32_t give_rank = __popcll(give_mask & ((1ull << lane) - 1ull));
                const uint32_t idle_rank = __popcll(idle_mask & ((1ull << lane) - 1ull));
                const bool giving = can_give && give_rank < n;
                const bool taking = !running && idle_rank < n;
                if (giving) {
                    donor_list[give_rank] = lane;
                    atomicAdd(&recs[4 * owner + 2], 1u);
                }
                __builtin_amdgcn_wave_barrier();                               // the takers read what the givers have just written
                uint32_t src = lane;
                if (taking) src = donor_list[idle_rank];
                const float ox = __shfl(lr.o.x, src), oy = __shfl(lr.o.y, src), oz = __shfl(lr.o.z, src);
                const float dx = __shfl(lr.d.x, src), dy = __shfl(lr.d.y, src), dz = __shfl(lr.d.z, src);
                const float s_limit = __shfl(limit, src), s_prune = __shfl(prune_abs, src), s_thr = __shfl(shadow_thr, src);
                const float s_idn = __shfl(inv_dn, src);
                const float ix = __shfl(inv.x, src), iy = __shfl(inv.y, src), iz = __shfl(inv.z, src);
                const uint32_t s_item = __shfl(item, src), s_gi = __shfl(gi, src), s_ii = __shfl(ii, src);
                const uint32_t s_owner = __shfl(owner, src), s_base = __shfl(base, src);
                if (taking) {                                                  // only `fast` walks are shared
                    lr.o = mk(ox, oy, oz); lr.d = mk(dx, dy, dz);
                    inv = mk(ix, iy, iz);
                    limit = s_limit; prune_abs = s_prune; shadow_thr = s_thr; inv_dn = s_idn;
                    item = s_item; gi = s_gi; ii = s_ii; owner = s_owner;
                    fast = true;
                    m_have = false; m_t = 0.0f; m_u = 0.0f; m_v = 0.0f; m_slot = 0;
                    cur_ref = wave_stack[s_base * RT_WAVE + src];           // an inner node: leaves are not given away
                    cur_t = __uint_as_float(wave_stack[(stack_lds + s_base) * RT_WAVE + src]);
                    sp = 0; base = 0;
                    running = true;
                }
                if (giving) ++base;
            }
        }
#ifdef RT_STATS
        const unsigned long long t_walk = clock64();
        if (serviced) { tm[9] += t_walk - t_round; }
#endif
        // ---- leaves: a lane that has reached a leaf keeps it until a fair share of the walking lanes has one
        // too, so that the triangle tests run on many lanes at once -------------------------------------------
        {
            const unsigned long long leaf_mask = __ballot(running && pending_leaf != RT_REF_NONE);
            const uint32_t n_leaf = __popcll(leaf_mask), n_run = __popcll(run_mask);
            if (n_leaf != 0 && (n_leaf >= RT_LEAF_BATCH || n_leaf * RT_LEAF_SHARE >= n_run)) {
#ifdef RT_STATS
                tm[5] += 1; tm[6] += n_leaf;
                const unsigned long long tl0 = clock64();
#endif
                if (running && pending_leaf != RT_REF_NONE) {
                    const uint32_t first = pending_leaf & RT_REF_SLOT_MASK;
                    const uint32_t cnt = (pending_leaf >> RT_REF_COUNT_SHIFT) & 7u;
                    pending_leaf = RT_REF_NONE;
                    n_triangle_tests += cnt;
                    bool improved = false;
                    // all of the leaf's triangles are fetched before the first test, so that the fetches overlap
                    // (entries past the leaf's count repeat its last triangle and are not tested)
                    // the leaf's triangles in two groups (three, then two): each group is fetched before its first test, so
                    // that the fetches overlap, without holding all five triangles in registers at once
#pragma unroll
                    for (uint32_t g0 = 0; g0 < RT_LEAF_TRIANGLES; g0 += 3) {
                        if (g0 >= cnt) break;
                        float4 h[3][3];
#pragma unroll
                        for (uint32_t k = 0; k < 3; ++k) {
                            if (g0 + k >= RT_LEAF_TRIANGLES) break;
                            const uint32_t slot = first + (g0 + k < cnt ? g0 + k : cnt - 1);
                            h[k][0] = S.tri_hot[3 * slot]; h[k][1] = S.tri_hot[3 * slot + 1]; h[k][2] = S.tri_hot[3 * slot + 2];
                        }
#pragma unroll
                        for (uint32_t k = 0; k < 3; ++k) {
                            if (g0 + k >= RT_LEAF_TRIANGLES) break;
                            const uint32_t slot = first + g0 + k;
                            float t, u, v;
                            if (g0 + k < cnt &&
                                triangle_tuv(mk(h[k][0].x, h[k][0].y, h[k][0].z), mk(h[k][0].w, h[k][1].x, h[k][1].y),
                                             mk(h[k][1].z, h[k][1].w, h[k][2].x), lr, t, u, v) &&
                                (!m_have || t < m_t || (t == m_t && slot < m_slot))) {
                                m_have = true; m_t = t; m_u = u; m_v = v; m_slot = slot;
                                improved = true;
                            }
                        }
                    }
                    if (improved) {
                        atomicMin((unsigned long long*)&recs[4 * owner], make_key(m_t, m_slot));
                        // does this candidate already prove a shadow ray blocked (see walk_item)?
                        if (m_t * inv_dn < shadow_thr) {
                            sp = base;
                            cur_ref = RT_REF_NONE;
                            recs[4 * owner + 3] = 1u;
                        }
                    }
                }
#ifdef RT_STATS
                tm[7] += clock64() - tl0;
#endif
            }
        }
        // ---- one entry per walking lane that holds no leaf ------------------------------------------------------
        if (running && pending_leaf == RT_REF_NONE) {
#ifdef RT_STATS
            ++dbg_steps;
#endif
            const uint32_t pair_t = __atomic_load_n(&recs[4 * owner + 1], __ATOMIC_RELAXED);     // closest hit any walker of the pair has
            const uint32_t pair_done = __atomic_load_n(&recs[4 * owner + 3], __ATOMIC_RELAXED);
            float cut = RT_INF;                                      // boxes entered beyond this cannot matter
            if (fast) {
                float lim = (m_have && m_t < limit) ? m_t : limit;
                if (pair_t != 0xFFFFFFFFu && __uint_as_float(pair_t) < lim) lim = __uint_as_float(pair_t);
                cut = lim * RT_PRUNE_REL + prune_abs;
            }
            uint32_t next_ref = RT_REF_NONE;                        // a child to go on with, kept off the stack
            float next_t = 0.0f;
            if (pair_done) {
                sp = base;                                           // another walker has proven the shadow
            } else if (cur_ref != RT_REF_NONE && !(cur_t > cut)) {
                if (cur_ref & RT_REF_LEAF_BIT) {
                    if ((cur_ref >> RT_REF_COUNT_SHIFT) & 7u) pending_leaf = cur_ref;
                } else {
                    // up to four entries, stored in the order the reference's walk would reach them
                    ++n_node_visits;
#ifdef RT_STATS
                    if (++dbg_walk_steps == 60 && level == RT_DEBUG_LONG_LEVEL) {
                        const uint32_t k = atomicAdd(&g_long_walks[0], 1u);
                        if (k < 64) {
                            unsigned int* o = g_long_walks + 1 + 24 * k;
                            o[0] = level; o[1] = item; o[2] = gi; o[3] = ii; o[4] = dbg_walk_steps;
                            const float4* q = L.fat_items + 4 * (size_t)item;
                            for (int j = 0; j < 4; ++j) { o[8 + 4 * j] = __float_as_uint(q[j].x); o[9 + 4 * j] = __float_as_uint(q[j].y); o[10 + 4 * j] = __float_as_uint(q[j].z); o[11 + 4 * j] = __float_as_uint(q[j].w); }
                        }
                    }
#endif
                    const float4* nq = S.nodes4 + 8 * (size_t)cur_ref;
                    const float4 nd0 = nq[0], nd1 = nq[1], nd2 = nq[2], nd3 = nq[3], nd4 = nq[4], nd5 = nq[5], nd6 = nq[6];
                    float t0, t1, t2, t3;
                    const bool h0 = slab(nd0.x, nd0.y, nd0.z, nd0.w, nd1.x, nd1.y, lr.o, inv, t0);
                    const bool h1 = slab(nd1.z, nd1.w, nd2.x, nd2.y, nd2.z, nd2.w, lr.o, inv, t1);
                    const bool h2 = slab(nd3.x, nd3.y, nd3.z, nd3.w, nd4.x, nd4.y, lr.o, inv, t2);
                    const bool h3 = slab(nd4.z, nd4.w, nd5.x, nd5.y, nd5.z, nd5.w, lr.o, inv, t3);
                    uint32_t r0 = (h0 && !(t0 > cut)) ? __float_as_uint(nd6.x) : RT_REF_NONE;
                    uint32_t r1 = (h1 && !(t1 > cut)) ? __float_as_uint(nd6.y) : RT_REF_NONE;
                    uint32_t r2 = (h2 && !(t2 > cut)) ? __float_as_uint(nd6.z) : RT_REF_NONE;
                    uint32_t r3 = (h3 && !(t3 > cut)) ? __float_as_uint(nd6.w) : RT_REF_NONE;
                    // on the fast path the nearest entry comes first (a sorting network over the entry distances; the
                    // order only decides how soon the pruning bites); otherwise the stored order is kept
                    if (fast) {
                        t0 = r0 == RT_REF_NONE ? RT_INF : t0; t1 = r1 == RT_REF_NONE ? RT_INF : t1;
                        t2 = r2 == RT_REF_NONE ? RT_INF : t2; t3 = r3 == RT_REF_NONE ? RT_INF : t3;
#define RT_CSWAP(ta, ra, tb, rb) { const bool sw = tb < ta; const float tt = sw ? ta : tb; ta = sw ? tb : ta; tb = tt; \
                                   const uint32_t rt_ = sw ? ra : rb; ra = sw ? rb : ra; rb = rt_; }
                        RT_CSWAP(t0, r0, t1, r1) RT_CSWAP(t2, r2, t3, r3) RT_CSWAP(t0, r0, t2, r2) RT_CSWAP(t1, r1, t3, r3) RT_CSWAP(t1, r1, t2, r2)
#undef RT_CSWAP
                    }
                    // go on with the first entry that passed, the others wait on the stack, the last one deepest
                    const uint32_t first = r0 != RT_REF_NONE ? 0u : r1 != RT_REF_NONE ? 1u : r2 != RT_REF_NONE ? 2u : r3 != RT_REF_NONE ? 3u : 4u;
                    if (r3 != RT_REF_NONE && first < 3u) { stack_push(sp, r3, t3); ++sp; }
                    if (r2 != RT_REF_NONE && first < 2u) { stack_push(sp, r2, t2); ++sp; }
                    if (r1 != RT_REF_NONE && first < 1u) { stack_push(sp, r1, t1); ++sp; }
                    next_ref = first == 0u ? r0 : first == 1u ? r1 : first == 2u ? r2 : first == 3u ? r3 : RT_REF_NONE;
                    next_t = first == 0u ? t0 : first == 1u ? t1 : first == 2u ? t2 : t3;
                }
            }
            // what this lane looks at next: the child chosen above, else the top of its stack
            if (next_ref == RT_REF_NONE && sp != base) {
                --sp;
                if (sp < stack_lds) {
                    next_ref = stack[sp * RT_WAVE];
                    next_t = __uint_as_float(stack_t[sp * RT_WAVE]);
                } else if (sp < stack_entries) {
                    const uint2 e = spill[(sp - stack_lds) * spill_lanes];
                    next_ref = e.x;
                    next_t = __uint_as_float(e.y);
                }                                                    // (else: an entry stack_push dropped)
            }
            cur_ref = next_ref; cur_t = next_t;
        }
        if (running && pending_leaf == RT_REF_NONE && cur_ref == RT_REF_NONE && sp == base) {
            running = false;                                         // settled in the next service phase
            pending = true;
        }
    }
    // work statistics: every wave owns a row of F.wave_stats (no atomics)
    {
        uint32_t a = n_node_visits, b = n_triangle_tests;
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
        if (lane == 0 && (a | b)) { F.wave_stats[(size_t)gwave * 8 + 5] += a; F.wave_stats[(size_t)gwave * 8 + 6] += b; }
    }
#ifdef RT_STATS
    tm[0] = clock64() - t_begin;
    tm[4] = wall_clock64();
    if (lane == 0 && gwave < 8192) for (int k = 0; k < 10; ++k) g_wave_timing[gwave * 10 + k] = tm[k];
#endif
}

// ======================================================================================================
// Kernel B as a packet walk: the wave walks the tree for all its pairs of one instance together while they agree on where
// to go, and lane by lane from where they part.  A wave takes 64 neighbouring walk items - at depth 0 the pairs of an
// 8 x 8 pixel block with an instance lie next to each other, at depth 1 the shadow rays of such a block's hits - and
// groups them by the tree they walk.  For a group the current node is wave-uniform: its 128 bytes arrive through scalar
// loads, once for all lanes, the four boxes are operands from scalar registers, each lane tests them for its own ray
// against its own distance bound, and a child is visited by the packet if enough lanes want it (`__ballot`); a child
// fewer lanes want goes onto those lanes' own stacks, and when the packet has nothing left every lane walks what it
// has collected on its own, as rt_traverse does.  The candidates a lane sees are those of its own walk in another
// order - a box is tested with the lane's own ray, a leaf's triangles only by the lanes whose ray passed the leaf's box -
// and the closest one by (t, slot) does not depend on the order (see rt_traverse).
// Which launches use it: DFrame::walk_packet_levels (the depths below that number).
// ======================================================================================================
#ifndef RT_PACKET_MIN
#define RT_PACKET_MIN 16           // lanes that have to want a node for the packet to visit it
#endif
#define RT_PACKET_ASIDE 32         // entries a lane may put aside before the wave walks them (DFrame::spill is sized for it)
// A read of scene data at a wave-uniform address, as a scalar load: the compiler only issues one when it knows the memory
// cannot change under the kernel, which the constant address space tells it (the scene arrays are written once, by
// rt_scene_create).
#if defined(__clang__) && defined(__HIP_DEVICE_COMPILE__) && !defined(RT_NO_SCALAR_LOADS)
typedef const float4 __attribute__((address_space(4))) RtUniformFloat4;
#define RT_UNIFORM_FLOAT4(p) ((RtUniformFloat4*)(p))
#else
#define RT_UNIFORM_FLOAT4(p) (p)
#endif

#ifdef RT_PACKET_WAVES
__attribute__((amdgpu_waves_per_eu(RT_PACKET_WAVES, RT_PACKET_WAVES)))
#endif
__global__ void __launch_bounds__(RT_BLOCK_THREADS) rt_traverse_packet(const DScene S, const DFrame F, const uint32_t level) {
    const LevelLists L = level_lists(F, level);
    {
        const uint32_t heavy_blocks = gridDim.x < RT_HEAVY_BLOCKS ? gridDim.x : RT_HEAVY_BLOCKS;
        if (blockIdx.x < heavy_blocks && *L.heavy_count != 0) traverse_heavy_pairs(S, F, level, heavy_blocks);
    }
    extern __shared__ uint32_t lds[];
    uint32_t* pre = lds;
    uint32_t* scratch = lds + F.n_seg + 1;
    const uint32_t n_items = build_prefix(L.fat_counts, F.n_seg, F.seg_fat, F.ovf_fat, pre, scratch);
    const uint32_t lane = threadIdx.x & (RT_WAVE - 1), wave = threadIdx.x / RT_WAVE, waves_per_block = blockDim.x / RT_WAVE;
    const uint32_t stack_lds = S.stack_lds, stack_entries = S.stack_entries;
    uint32_t* stack = lds + F.n_seg + 1 + blockDim.x + wave * (2 * stack_lds * RT_WAVE) + lane;       // this lane's own stack, as in rt_traverse
    uint32_t* stack_t = stack + stack_lds * RT_WAVE;
    uint2* spill = F.spill + (blockIdx.x * blockDim.x + threadIdx.x);
    const size_t spill_lanes = F.spill_lanes;
    // the packet's stack, one per wave: entry e = {reference, the lanes that want it (two words)}
    uint32_t* ustack = lds + F.n_seg + 1 + blockDim.x + waves_per_block * (2 * stack_lds * RT_WAVE) + wave * (3 * (stack_entries + 1));
    const uint32_t gwave = (blockIdx.x * blockDim.x + threadIdx.x) / RT_WAVE;
    const uint32_t n_waves = gridDim.x * waves_per_block;
    const uint32_t n_chunks = (n_items + RT_WAVE - 1) / RT_WAVE;
    uint32_t n_node_visits = 0, n_triangle_tests = 0;

    uint32_t sp = 0;
    // A lane's own stack holds what it has put aside during the packet - at most RT_PACKET_ASIDE entries, then the wave
    // stops to walk them - plus what a depth-first walk of one of them can need.
    const uint32_t own_entries = stack_entries + RT_PACKET_ASIDE;
    auto push_own = [&](uint32_t ref, float t) {
        if (sp < stack_lds) { stack[sp * RT_WAVE] = ref; stack_t[sp * RT_WAVE] = __float_as_uint(t); }
        else if (sp < own_entries) spill[(sp - stack_lds) * spill_lanes] = make_uint2(ref, __float_as_uint(t));
        else *F.overflow = F.overflow_seq;
        ++sp;
    };

    for (uint32_t c = gwave; c < n_chunks;) {
        const uint32_t v = c * RT_WAVE + lane;
        bool valid = v < n_items;
        const uint32_t fat = locate_run(pre, F.n_seg, F.seg_fat, c * RT_WAVE, lane, valid, lane);
        float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
        if (valid) { q0 = L.fat_items[4 * (size_t)fat]; q1 = L.fat_items[4 * (size_t)fat + 1]; q2 = L.fat_items[4 * (size_t)fat + 2]; q3 = L.fat_items[4 * (size_t)fat + 3]; }
        valid = valid && !(__float_as_uint(q3.y) & (RT_FAT_DONE | RT_FAT_HEAVY));
        Ray lr;
        lr.o = mk(q0.x, q0.y, q0.z);
        lr.d = mk(q1.x, q1.y, q1.z);
        const V3 inv = mk(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
        const float limit = q1.w, inv_dn = q2.x, shadow_thr = q2.y, prune_abs = q2.z;
        const uint32_t gi = __float_as_uint(q3.x), ii = __float_as_uint(q3.y) & RT_FAT_INSTANCE, root_ref = __float_as_uint(q2.w);
        const bool fast = (__float_as_uint(q3.y) & RT_FAT_FAST) != 0;
        bool m_have = false; float m_t = 0.0f, m_u = 0.0f, m_v = 0.0f; uint32_t m_slot = 0;
        bool proven = false;                                  // a shadow ray that is known to be blocked (see walk_item)
        sp = 0;

        // the candidate test of a leaf's triangle, as in rt_traverse
        auto test_triangle = [&](const float4 h0, const float4 h1, const float4 h2, uint32_t slot) {
            float t, u, w;
            if (triangle_tuv(mk(h0.x, h0.y, h0.z), mk(h0.w, h1.x, h1.y), mk(h1.z, h1.w, h2.x), lr, t, u, w) &&
                (!m_have || t < m_t || (t == m_t && slot < m_slot))) {
                m_have = true; m_t = t; m_u = u; m_v = w; m_slot = slot;
                if (m_t * inv_dn < shadow_thr) proven = true;
            }
        };
        auto current_cut = [&]() {
            if (!fast) return RT_INF;
            const float lim = (m_have && m_t < limit) ? m_t : limit;
            return lim * RT_PRUNE_REL + prune_abs;
        };

        // ---- lane by lane: what the lanes have put aside on their own stacks -------------------------------------------
        auto walk_own = [&]() {
            if (proven) sp = 0;                                         // (a blocked shadow ray needs nothing further)
            while (__ballot(sp != 0 && !proven)) {
                if (sp == 0 || proven) continue;
                --sp;
                uint32_t ref; float et;
                if (sp < stack_lds) { ref = stack[sp * RT_WAVE]; et = __uint_as_float(stack_t[sp * RT_WAVE]); }
                else if (sp < own_entries) { const uint2 e = spill[(sp - stack_lds) * spill_lanes]; ref = e.x; et = __uint_as_float(e.y); }
                else continue;                                                  // (an entry push_own dropped)
                const float cut = current_cut();
                if (et > cut) continue;
                if (ref & RT_REF_LEAF_BIT) {
                    const uint32_t first = ref & RT_REF_SLOT_MASK, cnt = (ref >> RT_REF_COUNT_SHIFT) & 7u;
                    n_triangle_tests += cnt;
                    for (uint32_t j = 0; j < cnt; ++j) {
                        const uint32_t slot = first + j;
                        test_triangle(S.tri_hot[3 * slot], S.tri_hot[3 * slot + 1], S.tri_hot[3 * slot + 2], slot);
                    }
                    continue;
                }
                ++n_node_visits;
                const float4* nq = S.nodes4 + 8 * (size_t)ref;
                const float4 nd0 = nq[0], nd1 = nq[1], nd2 = nq[2], nd3 = nq[3], nd4 = nq[4], nd5 = nq[5], nd6 = nq[6];
                float t0, t1, t2, t3;
                const bool h0 = slab(nd0.x, nd0.y, nd0.z, nd0.w, nd1.x, nd1.y, lr.o, inv, t0);
                const bool h1 = slab(nd1.z, nd1.w, nd2.x, nd2.y, nd2.z, nd2.w, lr.o, inv, t1);
                const bool h2 = slab(nd3.x, nd3.y, nd3.z, nd3.w, nd4.x, nd4.y, lr.o, inv, t2);
                const bool h3 = slab(nd4.z, nd4.w, nd5.x, nd5.y, nd5.z, nd5.w, lr.o, inv, t3);
                uint32_t r0 = (h0 && !(t0 > cut)) ? __float_as_uint(nd6.x) : RT_REF_NONE;
                uint32_t r1 = (h1 && !(t1 > cut)) ? __float_as_uint(nd6.y) : RT_REF_NONE;
                uint32_t r2 = (h2 && !(t2 > cut)) ? __float_as_uint(nd6.z) : RT_REF_NONE;
                uint32_t r3 = (h3 && !(t3 > cut)) ? __float_as_uint(nd6.w) : RT_REF_NONE;
                if (fast) {
                    t0 = r0 == RT_REF_NONE ? RT_INF : t0; t1 = r1 == RT_REF_NONE ? RT_INF : t1;
                    t2 = r2 == RT_REF_NONE ? RT_INF : t2; t3 = r3 == RT_REF_NONE ? RT_INF : t3;
#define RT_CSWAP(ta, ra, tb, rb) { const bool sw = tb < ta; const float tt = sw ? ta : tb; ta = sw ? tb : ta; tb = tt; \
                                       const uint32_t rt_ = sw ? ra : rb; ra = sw ? rb : ra; rb = rt_; }
                    RT_CSWAP(t0, r0, t1, r1) RT_CSWAP(t2, r2, t3, r3) RT_CSWAP(t0, r0, t2, r2) RT_CSWAP(t1, r1, t3, r3) RT_CSWAP(t1, r1, t2, r2)
#undef RT_CSWAP
                }
                if (r3 != RT_REF_NONE) push_own(r3, t3);                    // the last one deepest
                if (r2 != RT_REF_NONE) push_own(r2, t2);
                if (r1 != RT_REF_NONE) push_own(r1, t1);
                if (r0 != RT_REF_NONE) push_own(r0, t0);
            }
            if (proven) sp = 0;
        };

        // ---- the packets: one per tree among the chunk's pairs --------------------------------------------------------
        bool grouped = !valid;
        for (;;) {
            const unsigned long long todo = __ballot(!grouped);
            if (!todo) break;
            const uint32_t leader = __ffsll((long long)todo) - 1;
            const uint32_t root = __builtin_amdgcn_readfirstlane((uint32_t)__shfl((int)root_ref, (int)leader));
            const bool in_group = !grouped && root_ref == root;
            const unsigned long long gmask = __ballot(in_group);
            grouped = grouped || in_group;
            if ((uint32_t)__popcll(gmask) < RT_PACKET_MIN || root == RT_REF_NONE) {
                if (in_group && root != RT_REF_NONE) push_own(root_ref, q0.w);
                continue;
            }
            uint32_t usp = 0;
            if (lane == 0) { ustack[0] = root; ustack[1] = (uint32_t)gmask; ustack[2] = (uint32_t)(gmask >> 32); }
            usp = 1;
            __builtin_amdgcn_wave_barrier();
            while (usp) {
                --usp;
                const uint32_t ref = (uint32_t)__builtin_amdgcn_readfirstlane(ustack[3 * usp]);
                // (the builtin returns a signed int: without the casts a low word with its top bit set would spill over the high one)
                const unsigned long long m = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(ustack[3 * usp + 1]) |
                                             ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(ustack[3 * usp + 2]) << 32);
                const bool act = ((m >> lane) & 1ull) && !proven;
                if (ref & RT_REF_LEAF_BIT) {
                    const uint32_t first = ref & RT_REF_SLOT_MASK, cnt = (ref >> RT_REF_COUNT_SHIFT) & 7u;
                    if (act) n_triangle_tests += cnt;
                    for (uint32_t j = 0; j < cnt; ++j) {
                        const uint32_t slot = first + j;                                        // wave-uniform: scalar loads
                        const auto* hq = RT_UNIFORM_FLOAT4(S.tri_hot + 3 * (size_t)slot);
                        const float4 h0 = hq[0], h1 = hq[1], h2 = hq[2];
                        if (act) test_triangle(h0, h1, h2, slot);
                    }
                    continue;
                }
                // the node's 128 bytes through scalar loads (a wave-uniform address), once for all lanes
                const auto* nq = RT_UNIFORM_FLOAT4(S.nodes4 + 8 * (size_t)ref);
                const float4 nd0 = nq[0], nd1 = nq[1], nd2 = nq[2], nd3 = nq[3], nd4 = nq[4], nd5 = nq[5], nd6 = nq[6];
                if (act) ++n_node_visits;
                const float cut = current_cut();
                float t[4];
                bool pass[4];
                pass[0] = slab(nd0.x, nd0.y, nd0.z, nd0.w, nd1.x, nd1.y, lr.o, inv, t[0]);
                pass[1] = slab(nd1.z, nd1.w, nd2.x, nd2.y, nd2.z, nd2.w, lr.o, inv, t[1]);
                pass[2] = slab(nd3.x, nd3.y, nd3.z, nd3.w, nd4.x, nd4.y, lr.o, inv, t[2]);
                pass[3] = slab(nd4.z, nd4.w, nd5.x, nd5.y, nd5.z, nd5.w, lr.o, inv, t[3]);
                const uint32_t refs[4] = {__float_as_uint(nd6.x), __float_as_uint(nd6.y), __float_as_uint(nd6.z), __float_as_uint(nd6.w)};
                unsigned long long want[4];
                uint32_t near_key[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    pass[k] = act && refs[k] != RT_REF_NONE && pass[k] && !(t[k] > cut);
                    want[k] = __ballot(pass[k]);
                    // how near the child is, for the order of the visits: the smallest entry distance among the lanes that want
                    // it would be best; the first such lane's is what a packet can have for one lane read
                    const uint32_t who = want[k] ? (uint32_t)__ffsll((long long)want[k]) - 1 : 0u;
                    const float tk = __shfl(t[k], (int)who);
                    near_key[k] = want[k] ? (tk > 0.0f ? __builtin_amdgcn_readfirstlane(__float_as_uint(tk)) : 0u) : 0xFFFFFFFFu;
                }
                // far children first onto the stack, so that the nearest is popped first (the order only decides how soon the
                // lanes' distance bounds bite); a child few lanes want goes onto their own stacks
                uint32_t done_mask = 0;
#pragma unroll
                for (int round = 0; round < 4; ++round) {
                    int far = -1;
                    uint32_t far_key = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (!((done_mask >> k) & 1u) && want[k] && (far < 0 || near_key[k] >= far_key)) { far = k; far_key = near_key[k]; }
                    if (far < 0) break;
                    done_mask |= 1u << far;
                    const unsigned long long wm = far == 0 ? want[0] : far == 1 ? want[1] : far == 2 ? want[2] : want[3];
                    const uint32_t wr = far == 0 ? refs[0] : far == 1 ? refs[1] : far == 2 ? refs[2] : refs[3];
                    if ((uint32_t)__popcll(wm) >= RT_PACKET_MIN) {
                        if (usp <= stack_entries) {
                            if (lane == 0) { ustack[3 * usp] = wr; ustack[3 * usp + 1] = (uint32_t)wm; ustack[3 * usp + 2] = (uint32_t)(wm >> 32); }
                            ++usp;
                        } else *F.overflow = F.overflow_seq;
                    } else {
                        const bool mine = far == 0 ? pass[0] : far == 1 ? pass[1] : far == 2 ? pass[2] : pass[3];
                        const float mt = far == 0 ? t[0] : far == 1 ? t[1] : far == 2 ? t[2] : t[3];
                        if (mine) push_own(wr, mt);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (__ballot(sp + 4 > RT_PACKET_ASIDE)) walk_own();     // (every lane's stack is empty afterwards)
            }
        }
        walk_own();

        // ---- the pairs' closest hits join their rays' other candidates (scene.rs:133-136), as in rt_traverse ---------------
        if (valid && m_have) {
            const float tw = world_t(S, S.instances[ii], lr, load_any_origin(S, F, level, gi), m_t, m_slot);
            L.item_results[fat] = make_float4(m_t, m_u, m_v, __uint_as_float(m_slot));
            atomicMin(&L.best[gi], make_key(tw, RT_KEY_MESH | fat));
        }
        // the next chunk: the wave's own first, further ones from the depth's counter
        uint32_t taken = 0;
        if (lane == 0) taken = atomicAdd(&F.chunk_counter[level], 1u);
        c = n_waves + __builtin_amdgcn_readfirstlane(taken);
    }
    {
        uint32_t a = n_node_visits, b = n_triangle_tests;
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
        if (lane == 0 && (a | b)) { F.wave_stats[(size_t)gwave * 8 + 5] += a; F.wave_stats[(size_t)gwave * 8 + 6] += b; }
    }
}

// ======================================================================================================
// Second half of kernel C: the Phong sum of shade_phong / radiance (raytracer.rs:75-84, 137-153) for the hits
// of the previous depth, now that their shadow rays have been traced.  Block-wide (it builds a prefix table).
// ======================================================================================================
// rt_shade with setup_next: a slot in a shared overflow segment is not in the block's own segment, so the block keeps a
// list of those it took in device memory (DFrame::extras; as long as its own segments, since what a block emits varies
// with what its chunks show: on the 1080p bunnies frame the fullest segment of child rays holds 2.3 times the mean).  A
// block that takes more than the list holds spoils the frame, which is then redone with longer lists, like any frame
// whose lists were too short.
// Its entry stays in the list, though, and nobody sets it up: its `keys_per_slot` keys (the key of a ray; the keys of the
// shadow rays of a hit record) are written here, so that the next depth finds well-formed ones.
__device__ __forceinline__ void note_extra(const DFrame& F, uint32_t slot, uint32_t seg_cap, uint32_t* list, uint32_t list_len, uint32_t* n,
                                           unsigned long long* keys, uint32_t keys_per_slot) {
    if (slot == RT_REF_NONE || slot < (F.n_seg - 1) * seg_cap) return;
    const uint32_t k = atomicAdd(n, 1u);
    if (k < list_len) { list[k] = slot; return; }
    *F.overflow = F.overflow_seq;
    for (uint32_t j = 0; j < keys_per_slot; ++j) keys[(size_t)slot * keys_per_slot + j] = RT_KEY_NONE;
}

__device__ void phong_pass(const DScene& S, const DFrame& F, const uint32_t level, uint32_t* pre, uint32_t* scratch) {
    const unsigned int* hit_counts = level_lists(F, level).hit_counts;
    const unsigned long long* shadow_keys = level_lists(F, level).best + F.shadow_base;      // see DFrame::rays
    const uint32_t q = (level & 1) ^ 1;            // parity of the depth whose hits are finished here
    const uint32_t n_hits = build_prefix(hit_counts, F.n_seg, F.seg_hits, F.ovf_hits, pre, scratch);
    const uint32_t lane = threadIdx.x & (RT_WAVE - 1);
    const uint32_t gwave = (blockIdx.x * blockDim.x + threadIdx.x) / RT_WAVE;
    const uint32_t n_waves = gridDim.x * (blockDim.x / RT_WAVE);
    for (uint32_t c = gwave; (unsigned long long)c * RT_WAVE < n_hits; c += n_waves) {
        const uint32_t v = c * RT_WAVE + lane;
        const uint32_t hs = locate_chunk(pre, F.n_seg, F.seg_hits, c * RT_WAVE, n_hits, lane);
        if (v >= n_hits) continue;
        const float4* H = F.hits[q] + 3 * (size_t)hs;
        const float4 h0 = H[0], h1 = H[1], h2 = H[2];
        const V3 position = mk(h0.x, h0.y, h0.z), n = mk(h1.x, h1.y, h1.z), view = mk(h2.x, h2.y, h2.z);
        const RtMaterial& mat = S.materials[__float_as_uint(h0.w)];
        const float throughput = h1.w;
        const uint32_t pw = __float_as_uint(h2.w), pixel = pw & RT_META_PIXEL_MASK;
        V3 color = mk(0, 0, 0);
        uint32_t shadow_bits = 0;
        for (uint32_t li = 0; li < S.n_lights; ++li) {
            const DLight& L = S.lights[li];
            const bool directional = L.pos[3] == 0.0f;
            // trace_shadow_ray, raytracer.rs:171-187: any hit towards a directional light, a hit closer than a point light
            const unsigned long long key = shadow_keys[hs * S.n_lights + li];
            const bool is_shadow = key != RT_KEY_NONE &&
                                   (directional || __uint_as_float((uint32_t)(key >> 32)) < distance3(position, ld3(L.world)));
            if (is_shadow && li < 4) shadow_bits |= RT_FLAG_SHADOW_LIGHT0 << li;
            V3 l = normalize3(directional ? mk(L.pos[0], L.pos[1], L.pos[2]) : ld3(L.world) - position);   // raytracer.rs:209-220
            float l_dot_n, r_dot_v;
            phong_cosines(l, n, view, l_dot_n, r_dot_v);
            V3 emissive = ld3(mat.emissive);
            V3 ambient = cmul(ld3(L.ambient), ld3(mat.ambient));
            V3 zero = mk(0.0f, 0.0f, 0.0f);
            V3 diffuse = is_shadow ? zero : cmul(ld3(L.diffuse), ld3(mat.diffuse)) * l_dot_n;
            V3 specular = is_shadow ? zero : cmul(ld3(L.specular), ld3(mat.specular)) * powf(r_dot_v, mat.shininess);
            V3 rad = emissive + ambient + diffuse + specular;
            color = (li == 0) ? rad : color + rad;                     // color::add_option
        }
        add_color(F, pixel, color * throughput);
        if (F.aux_flags && (pw & RT_HIT_DEPTH0) && shadow_bits) atomicOr(F.aux_flags + pixel, shadow_bits);
    }
}

// ======================================================================================================
// Kernel C: per ray of this depth, what raytrace_impl / shade do once the closest hit is known
// (raytracer.rs:65-135), except the Phong sum, which has to wait for its shadow rays (phong_pass, one depth later).
// Shadow rays only record their verdict (raytracer.rs:171-187).
// ======================================================================================================
// six waves per SIMD, six blocks per compute unit: left alone the kernel takes 88 registers (five waves); held to 80 it spills
// four and still gains (0.796 -> 0.757 ms per queued 1080p bunnies frame; four waves without any spill: 1.15 ms)
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 6
#endif
__attribute__((amdgpu_waves_per_eu(RT_SHADE_WAVES, RT_SHADE_WAVES)))
__global__ void __launch_bounds__(RT_SHADE_THREADS) rt_shade(const DScene S, const DFrame F, const uint32_t level, const uint32_t setup_next) {
#ifdef RT_STATS
    unsigned long long ts[8];
    ts[0] = clock64();
#endif
    const LevelLists L = level_lists(F, level);
    extern __shared__ uint32_t lds[];
    // With setup_next the block keeps the slots it takes from the shared overflow segments; what setup_rays needs
    // later (staged primitives, range-tree stacks, rings) lies over the prefix table, which is done with by then.
    uint32_t* extra_rays = F.extras + (size_t)blockIdx.x * (F.extra_rays_cap + F.extra_hits_cap);
    uint32_t* extra_hits = extra_rays + F.extra_rays_cap;
    uint32_t* pre = lds;
    uint32_t* scratch = pre + F.n_seg + 1;
    __shared__ uint32_t out_rays, out_hits, n_extra_rays, n_extra_hits, fat_n, slow_n;
    if (threadIdx.x == 0) { out_rays = 0; out_hits = 0; n_extra_rays = 0; n_extra_hits = 0; fat_n = 0; slow_n = 0; }
    const uint32_t p = level & 1, q = p ^ 1;
    unsigned long long* next_best = F.best[q];
    const uint32_t n_rays = build_prefix(L.ray_counts, F.n_seg, F.seg_rays, F.ovf_rays, pre, scratch);
    const float4* rays = F.rays[p];
    float4* rays_out = F.rays[q];
    const uint32_t lane = threadIdx.x & (RT_WAVE - 1);
    const uint32_t gwave = (blockIdx.x * blockDim.x + threadIdx.x) / RT_WAVE;
    const uint32_t n_waves = gridDim.x * (blockDim.x / RT_WAVE);
    uint32_t n_primary = 0, n_shadow = 0, n_reflected = 0, n_transmitted = 0;
#ifdef RT_STATS
    ts[1] = clock64();
#endif

    for (uint32_t c = gwave; (unsigned long long)c * RT_WAVE < n_rays; c += n_waves) {
        const uint32_t v = c * RT_WAVE + lane;
        const bool active = v < n_rays;
        uint32_t gi = 0;
        RayRec cur;
        cur.ray.o = cur.ray.d = mk(0, 0, 0);
        cur.throughput = 0.0f;
        cur.meta = 0;
        unsigned long long key = RT_KEY_NONE;
        {
            const uint32_t at = locate_chunk(pre, F.n_seg, F.seg_rays, c * RT_WAVE, n_rays, lane);
            if (active) { gi = at; cur = load_rec(rays, gi); key = L.best[gi]; }
        }
        const float best_t = __uint_as_float((uint32_t)(key >> 32));
        const uint32_t order = (uint32_t)key;
        // A frame whose lists overflowed (it is invalid and will be redone) may hold keys nobody wrote: a key that names no
        // walk item or primitive counts as a miss, so that every index stays inside its array.
        bool have = key != RT_KEY_NONE;
        if (order & RT_KEY_MESH) have = have && (order & ~RT_KEY_MESH) < F.n_fat_slots;
        else have = have && (order & 0x1FFFFFFFu) < ((order >> 29) == RT_PRIM_PLANE ? S.n_planes : (order >> 29) == RT_PRIM_SPHERE ? S.n_spheres :
                                                    (order >> 29) == RT_PRIM_TRIANGLE ? S.n_triangles : 0u);

        bool emit_r = false, emit_t = false, emit_hit = false;
        RayRec rec_r, rec_t;
        rec_r.ray.o = rec_r.ray.d = rec_t.ray.o = rec_t.ray.d = mk(0, 0, 0);
        rec_r.throughput = rec_t.throughput = 0.0f;
        rec_r.meta = rec_t.meta = 0;
        Hitpoint hit;
        hit.position = hit.hit_normal = hit.position_for_refraction = mk(0, 0, 0);
        hit.on_frontside = false;
        hit.material = 0;
        const uint32_t pixel = cur.meta & RT_META_PIXEL_MASK;

        if (active) {
            // ---- primary / reflected / transmitted ray (shadow rays are not in this list; phong_pass reads their
            // closest candidates itself) ----------------------------------------------------------------------
            const uint32_t kind = (cur.meta >> RT_META_KIND_SHIFT) & 1u;
            if (level == 0) ++n_primary; else if (kind) ++n_transmitted; else ++n_reflected;
            if (!have) {
                // raytrace returned None: the caller substitutes the background (main.rs:58-61, raytracer.rs:92,110,112)
                add_color(F, pixel, background(S, cur.ray.d) * cur.throughput);
            } else if (F.mode == RT_MODE_DEPTH_MAP) {
                // Raytracer::depth_map, raytracer.rs:21-28 (DEPTH_MAP_EXP_BASE 2, DEPTH_MAP_BRIGHTNESS_SCALE 1.5)
                const float brightness = powf(2.0f, -best_t) * 1.5f;
                add_color(F, pixel, mk(brightness, brightness, brightness) * cur.throughput);
            } else {
                Closest cl;
                cl.have = true; cl.t = best_t; cl.inst = RT_INSTANCE_NONE; cl.slot = 0; cl.tl = cl.u = cl.v = 0.0f;
                if (order & RT_KEY_MESH) {
                    const uint32_t item = order & ~RT_KEY_MESH;
                    float4 res = L.item_results[item];
                    cl.inst = __float_as_uint(L.fat_items[4 * (size_t)item + 3].y) & RT_FAT_INSTANCE;
                    if (__float_as_uint(res.w) >= S.n_tri_slots) res.w = 0.0f;         // (stale results of such a frame, see above)
                    if (cl.inst >= S.n_instances) cl.inst = 0;
                    cl.tl = res.x; cl.u = res.y; cl.v = res.z; cl.slot = __float_as_uint(res.w);
                    cl.prim = ((uint32_t)RT_PRIM_MESH_TRIANGLE << 30) | __float_as_uint(S.tri_cold[3 * cl.slot].w);
                } else {
                    cl.prim = ((order >> 29) << 30) | (order & 0x1FFFFFFFu);
                }
                hit = finalize_hit(S, cur.ray, cl);
                const RtMaterial& mat = S.materials[hit.material];
                const uint32_t mtype = mat.material_type;
                const bool returns_some = !(mtype == RT_MATERIAL_PHONG && S.n_lights == 0);
                if (F.aux_flags) {
                    if (level == 0) {
                        F.aux_prim[pixel] = cl.prim;
                        F.aux_inst[pixel] = cl.inst;
                        F.aux_t[pixel] = best_t;
                        atomicOr(F.aux_flags + pixel, RT_FLAG_HIT | (hit.on_frontside ? RT_FLAG_FRONTSIDE : 0u));
                    } else if (level == 1 && returns_some) {
                        atomicOr(F.aux_flags + pixel, kind ? RT_FLAG_REFRACTION_HIT : RT_FLAG_REFLECTION_HIT);
                    }
                }
                if (mtype == RT_MATERIAL_REFLECT_AND_REFRACT) {
                    // raytracer.rs:96-123
                    V3 V = -cur.ray.d;
                    float n1 = hit.on_frontside ? mat.index_outer : mat.index_inner;
                    float n2 = hit.on_frontside ? mat.index_inner : mat.index_outer;
                    V3 td = normalize3(transmitted_dir(V, hit.hit_normal, n1, n2));
                    V3 rd = normalize3(reflected_dir(V, hit.hit_normal));
                    float k_reflected = fresnel_reflection(rd, td, hit.hit_normal, n1, n2);
                    float k_transmitted = 1.0f - k_reflected;
                    rec_r.ray.o = hit.position; rec_r.ray.d = rd;
                    rec_r.throughput = cur.throughput * k_reflected;
                    rec_r.meta = pixel;
                    rec_t.ray.o = hit.position_for_refraction; rec_t.ray.d = td;
                    rec_t.throughput = cur.throughput * k_transmitted;
                    rec_t.meta = pixel | (1u << RT_META_KIND_SHIFT);
                    if (level + 1 < F.max_depth) {
                        emit_r = emit_t = true;
                    } else {
                        // raytrace_impl returns None without tracing (raytracer.rs:66): background of each child
                        add_color(F, pixel, background(S, rd) * rec_r.throughput);
                        add_color(F, pixel, background(S, td) * rec_t.throughput);
                    }
                } else {
                    if (S.n_lights > 0) {
                        emit_hit = true;                                   // shade_phong, finished by phong_pass
                    } else if (mtype == RT_MATERIAL_PHONG) {
                        add_color(F, pixel, background(S, cur.ray.d) * cur.throughput);      // shade() returned None: as a miss
                    }
                    if (mtype == RT_MATERIAL_REFLECT_AND_PHONG) {
                        // raytracer.rs:86-94
                        V3 rd = normalize3(reflected_dir(-cur.ray.d, hit.hit_normal));
                        rec_r.ray.o = hit.position; rec_r.ray.d = rd;
                        rec_r.throughput = cur.throughput * 0.8f;          // REFLECTION_DIM_FACTOR
                        rec_r.meta = pixel;
                        if (level + 1 < F.max_depth) emit_r = true;
                        else add_color(F, pixel, background(S, rd) * rec_r.throughput);
                    }
                }
            }
        }

        // ---- append: hit record, its shadow rays (one per light, raytracer.rs:77-79), child rays ----------
        const uint32_t hit_slot = append_slot(emit_hit, &out_hits, F.seg_hits, F.n_seg, L.next_hit_counts, F.ovf_hits, F.overflow, F.overflow_seq, lane);
        if (setup_next) note_extra(F, hit_slot, F.seg_hits, extra_hits, F.extra_hits_cap, &n_extra_hits, next_best + F.shadow_base, S.n_lights);
        if (hit_slot != RT_REF_NONE) {
            float4* H = F.hits[p] + 3 * (size_t)hit_slot;
            H[0] = make_float4(hit.position.x, hit.position.y, hit.position.z, __uint_as_float(hit.material));
            H[1] = make_float4(hit.hit_normal.x, hit.hit_normal.y, hit.hit_normal.z, cur.throughput);
            H[2] = make_float4(-cur.ray.d.x, -cur.ray.d.y, -cur.ray.d.z, __uint_as_float(pixel | (level == 0 ? RT_HIT_DEPTH0 : 0u)));
        }
        // one shadow ray per light (raytracer.rs:77-79), traced at the next depth straight from the hit record (shadow_rec)
        if (hit_slot != RT_REF_NONE) n_shadow += S.n_lights;
        {
            const uint32_t slot = append_slot(emit_r, &out_rays, F.seg_rays, F.n_seg, L.next_ray_counts, F.ovf_rays, F.overflow, F.overflow_seq, lane);
            if (setup_next) note_extra(F, slot, F.seg_rays, extra_rays, F.extra_rays_cap, &n_extra_rays, next_best, 1);
            if (slot != RT_REF_NONE) store_rec(rays_out, slot, rec_r);
        }
        {
            const uint32_t slot = append_slot(emit_t, &out_rays, F.seg_rays, F.n_seg, L.next_ray_counts, F.ovf_rays, F.overflow, F.overflow_seq, lane);
            if (setup_next) note_extra(F, slot, F.seg_rays, extra_rays, F.extra_rays_cap, &n_extra_rays, next_best, 1);
            if (slot != RT_REF_NONE) store_rec(rays_out, slot, rec_t);
        }
    }

#ifdef RT_STATS
    ts[2] = clock64();
#endif
    // ---- ray statistics: every wave owns a row of F.wave_stats (no atomics) ----------------------------
    {
        uint32_t vals[4] = {n_primary, n_shadow, n_reflected, n_transmitted};      // columns 1..4 (column 0: setup_rays)
        for (int k = 0; k < 4; ++k) {
            uint32_t x = vals[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            if (lane == 0 && x) F.wave_stats[(size_t)gwave * 8 + 1 + k] += x;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        L.next_ray_counts[blockIdx.x] = out_rays < F.seg_rays ? out_rays : F.seg_rays;
        L.next_hit_counts[blockIdx.x] = out_hits < F.seg_hits ? out_hits : F.seg_hits;
    }
    // the hits of the depth above are complete now that this depth's rays - their shadow rays among them - are traced
#ifdef RT_STATS
    ts[3] = clock64();
#endif
    if (level > 0 && S.n_lights > 0) phong_pass(S, F, level, pre, scratch);
#ifdef RT_STATS
    ts[4] = clock64();
#endif
    // ---- the next depth's rt_setup, block by block: the rays and hit records this block has just appended are all it needs,
    // and a block is the only writer of its segments, so no other block has to be waited for
    if (setup_next) {
        // The block's own stores to device memory have to be done before its other waves load them.  A fence of workgroup
        // scope waits for exactly that (__syncthreads alone only orders LDS); a device-wide __threadfence() would write back
        // and invalidate the whole L2 of the XCD, about 100 us per launch.
        __threadfence_block();
        __syncthreads();
        SetupInput in;
        in.pre = in.pre_hits = nullptr;
        in.table = in.table_scratch = nullptr;
        in.own_rays = out_rays < F.seg_rays ? out_rays : F.seg_rays;
        in.own_hits = out_hits < F.seg_hits ? out_hits : F.seg_hits;
        in.extra_rays = extra_rays; in.extra_hits = extra_hits;
        in.n_rays = in.own_rays + (n_extra_rays < F.extra_rays_cap ? n_extra_rays : F.extra_rays_cap);
        in.n_hits = in.own_hits + (n_extra_hits < F.extra_hits_cap ? n_extra_hits : F.extra_hits_cap);
        uint32_t* stage = pre;                                        // every wave has left phong_pass (barrier above)
        stage_analytic(S, stage);
        setup_rays<true>(S, F, level + 1, in, stage, &fat_n, &slow_n);
    }
#ifdef RT_STATS
    ts[5] = clock64();
    if (lane == 0 && gwave < 8192) for (int k = 0; k < 6; ++k) g_shade_timing[gwave * 8 + k] = ts[k];
#endif
}

// color.rs:42-52 + canvas layout.  One thread per local pixel.
//  tiled == 0: row-major frame (bytes_per_pixel * width * height); padding pixels are skipped
//  tiled == 1: this shard's tiles back to back, pixels row-major inside a tile, padding written as 0
__global__ void rt_resolve(const DFrame F, uint8_t* out, uint32_t bytes_per_pixel, uint32_t tiled) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= F.n_pixels) return;
    uint32_t view = 0, col = 0, row = 0;
    bool inside = pixel_coords(F, p, view, col, row);
    uint8_t rgb[3] = {0, 0, 0};
    if (inside) {
        unsigned int poison = F.poison[p];
        for (int c = 0; c < 3; ++c) {
            double sum = (double)F.acc[(size_t)c * F.n_pixels + p] * (1.0 / RT_FIXED_ONE);
            if (poison & RT_POISON_BIG) {
                sum += F.acc_big[(size_t)c * F.n_pixels + p];
                F.acc_big[(size_t)c * F.n_pixels + p] = 0.0;        // only pixels that used it clear it: no per-frame clear
            }
            float v = (float)sum;
            bool is_nan = (poison >> c) & 1u, pinf = (poison >> (3 + c)) & 1u, ninf = (poison >> (6 + c)) & 1u;
            if (is_nan || (pinf && ninf)) v = __uint_as_float(0x7FC00000u);
            else if (pinf) v = __uint_as_float(0x7F800000u);
            else if (ninf) v = __uint_as_float(0xFF800000u);
            float clamped = (v > 0.0f) ? ((v < 1.0f) ? v : 1.0f) : 0.0f;     // glm::clamp, NaN -> 0
            rgb[c] = (uint8_t)(int)(clamped * 255.0f);                        // `as u8` truncates
        }
    }
    size_t at;                                                      // view v starts at v * (bytes of one shard or frame)
    if (tiled) {
        uint32_t k = p / (RT_TILE_DIM * RT_TILE_DIM), q = p % (RT_TILE_DIM * RT_TILE_DIM);      // k counts across views
        uint32_t b = q / 64, l = q % 64;
        uint32_t px = (b % 4) * 8 + (l % 8), py = (b / 4) * 8 + (l / 8);
        at = ((size_t)k * RT_TILE_DIM * RT_TILE_DIM + py * RT_TILE_DIM + px) * bytes_per_pixel;
    } else {
        if (!inside) return;
        at = (((size_t)view * F.height + row) * F.width + col) * bytes_per_pixel;
    }
    out[at] = rgb[0]; out[at + 1] = rgb[1]; out[at + 2] = rgb[2];
    if (bytes_per_pixel == 4) out[at + 3] = inside ? 255 : 0;
}

// Scatters the aux planes (local pixel order) into row-major W*H arrays.
__global__ void rt_resolve_aux(const DFrame F, uint32_t* prim, uint32_t* inst, uint8_t* flags, float* hit_t) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= F.n_pixels) return;
    uint32_t view, col, row;
    if (!pixel_coords(F, p, view, col, row)) return;
    size_t at = ((size_t)view * F.height + row) * F.width + col;
    unsigned int fl = F.aux_flags[p];
    bool hit = fl & RT_FLAG_HIT;
    if (prim) prim[at] = hit ? F.aux_prim[p] : RT_PRIM_NONE;
    if (inst) inst[at] = hit ? F.aux_inst[p] : RT_INSTANCE_NONE;
    if (flags) flags[at] = (uint8_t)fl;
    if (hit_t) hit_t[at] = hit ? F.aux_t[p] : __uint_as_float(0x7FC00000u);
}

// Gathered shard buffers (rank-major; inside a rank view-major, then tile-major) -> row-major frames, view after view.
// One thread per pixel of every frame.
__global__ void rt_assemble(const uint8_t* gathered, uint8_t* frame, uint32_t width, uint32_t height, uint32_t tiles_x,
                            uint32_t shard_count, uint32_t tiles_per_shard, uint32_t bytes_per_pixel, uint32_t n_views) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)width * height * n_views) return;
    const uint32_t view = (uint32_t)(i / ((size_t)width * height)), j = (uint32_t)(i % ((size_t)width * height));
    uint32_t row = j / width, col = j % width;
    uint32_t tile = (row / RT_TILE_DIM) * tiles_x + col / RT_TILE_DIM;
    uint32_t shard = tile % shard_count, k = tile / shard_count;
    size_t src = ((((size_t)shard * n_views + view) * tiles_per_shard + k) * RT_TILE_DIM * RT_TILE_DIM +
                  (row % RT_TILE_DIM) * RT_TILE_DIM + col % RT_TILE_DIM) * bytes_per_pixel;
    size_t dst = i * bytes_per_pixel;
    for (uint32_t c = 0; c < bytes_per_pixel; ++c) frame[dst + c] = gathered[src + c];
}

// ---- launch wrappers (called from rt_api.hip) ----------------------------------------------------

static size_t list_lds_bytes(uint32_t n_seg) { return (size_t)(n_seg + 1 + RT_MAX_LIST_THREADS) * sizeof(uint32_t); }

static size_t traverse_lds_bytes(uint32_t n_seg, uint32_t stack_entries) {
    // prefix table and scan scratch; two words per stack entry (child reference, box entry distance), one column
    // per lane; then, 16-byte aligned, one 16-byte pair record per lane
    return (size_t)(n_seg + 1 + RT_BLOCK_THREADS) * sizeof(uint32_t) +
           (size_t)(RT_BLOCK_THREADS / RT_WAVE) * 2 * stack_entries * RT_WAVE * sizeof(uint32_t) + 16 + (size_t)RT_BLOCK_THREADS * 16;
}

static size_t packet_lds_bytes(uint32_t n_seg, uint32_t stack_lds, uint32_t stack_entries) {
    // as the walk's, without the pair records, plus the packets' stacks (three words per entry, one stack per wave)
    return (size_t)(n_seg + 1 + RT_BLOCK_THREADS) * sizeof(uint32_t) +
           (size_t)(RT_BLOCK_THREADS / RT_WAVE) * (2 * stack_lds * RT_WAVE + 3 * (stack_entries + 1)) * sizeof(uint32_t);
}

// Resident blocks of the traversal kernel per compute unit (registers and the LDS stacks decide).
extern "C" int rt_query_traverse_blocks_per_cu(uint32_t n_seg, uint32_t stack_entries) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, rt_traverse, RT_BLOCK_THREADS, traverse_lds_bytes(n_seg, stack_entries)) != hipSuccess)
        return -1;
    return n;
}

// `matrices`: F->n_views column-major screen_to_world matrices in host memory (copied into the launches' arguments).
extern "C" hipError_t rt_launch_generate(const DFrame* F, const float* matrices, uint32_t blocks, hipStream_t stream) {
    for (uint32_t first = 0; first < F->n_views; first += RT_VIEWS_PER_LAUNCH) {
        const uint32_t n = F->n_views - first < RT_VIEWS_PER_LAUNCH ? F->n_views - first : RT_VIEWS_PER_LAUNCH;
        DViews V;
        for (uint32_t k = 0; k < RT_VIEWS_PER_LAUNCH; ++k)                    // (the unused ones repeat the first)
            for (uint32_t i = 0; i < 16; ++i) V.m[k][i] = matrices[(size_t)16 * (first + (k < n ? k : 0)) + i];
        hipLaunchKernelGGL(rt_generate, dim3(blocks), dim3(RT_BLOCK_THREADS), 0, stream, *F, V, first * F->view_pixels, n * F->view_pixels);
    }
    return hipGetLastError();
}

// One recursion depth: A, B, C and (from depth 1 on) D.  `blocks` is the grid of the list-producing
// kernels (F->n_seg - 1), `traverse_blocks` that of the BVH walk.
// `marks`, when not null, are four events recorded before A and after A', B and C (kernel timing mode).
// With F->fuse_setup, rt_shade also does the next depth's rt_setup (every depth but `last`), and rt_setup itself is only
// launched at depth 0.
// Blocks of rt_setup a compute unit holds at once with lists of n_seg segments: its grid is kept to what is resident in
// one go (block b fills list segment b, so any grid up to n_seg - 1 works; the segments of blocks that are not launched stay empty).
extern "C" int rt_query_setup_blocks_per_cu(const DScene* S, uint32_t n_seg) {
    const size_t setup_words = analytic_stage_words(*S) + setup_work_words(S->range_depth, RT_SETUP_THREADS);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, rt_setup, RT_SETUP_THREADS,
                                                     list_lds_bytes(n_seg) + setup_words * sizeof(uint32_t)) != hipSuccess)
        return -1;
    return n;
}

extern "C" hipError_t rt_launch_level(const DScene* S, const DFrame* F, uint32_t level, uint32_t last, uint32_t blocks, uint32_t setup_blocks,
                                      uint32_t traverse_blocks, hipStream_t stream, hipEvent_t* marks) {
    const size_t lds = list_lds_bytes(F->n_seg);
#ifdef RT_STATS
    const char* stop = getenv("RT_DEBUG_STOP_LEVEL");           // keep the lists of that depth as rt_traverse left them
    if (stop && level > (uint32_t)atoi(stop)) return hipSuccess;
#endif
    if (marks) (void)hipEventRecord(marks[0], stream);
    const size_t setup_words = analytic_stage_words(*S) + setup_work_words(S->range_depth, RT_SETUP_THREADS);
    if (level == 0 || !F->fuse_setup)
        hipLaunchKernelGGL(rt_setup, dim3(setup_blocks < blocks ? setup_blocks : blocks), dim3(RT_SETUP_THREADS),
                           lds + setup_words * sizeof(uint32_t), stream, *S, *F, level);
    // In a scene whose instances all have culling boxes only stray rays (NaN components) reach this kernel, if any:
    // a small grid keeps the usual empty launch cheap.  (Block b continues list segment b, so any grid <= blocks works.)
    const uint32_t slow_blocks = (S->all_cullable && blocks > 64) ? 64 : blocks;
    if (F->slow_launched)
        hipLaunchKernelGGL(rt_setup_slow, dim3(slow_blocks), dim3(RT_BLOCK_THREADS), lds + analytic_stage_words(*S) * sizeof(uint32_t), stream,
                           *S, *F, level);
    if (marks) (void)hipEventRecord(marks[1], stream);
    if (S->n_instances > 0 && level < F->walk_packet_levels)
        hipLaunchKernelGGL(rt_traverse_packet, dim3(traverse_blocks), dim3(RT_BLOCK_THREADS),
                           packet_lds_bytes(F->n_seg, S->stack_lds, S->stack_entries), stream, *S, *F, level);
    else if (S->n_instances > 0)
        hipLaunchKernelGGL(rt_traverse, dim3(traverse_blocks), dim3(RT_BLOCK_THREADS), traverse_lds_bytes(F->n_seg, S->stack_lds),
                           stream, *S, *F, level);
#ifdef RT_STATS
    if (stop && level == (uint32_t)atoi(stop)) return hipGetLastError();
#endif
    if (marks) (void)hipEventRecord(marks[2], stream);
    const uint32_t setup_next = (F->fuse_setup && level < last) ? 1u : 0u;
    const size_t shade_lds = setup_next ? std::max(lds, setup_words * sizeof(uint32_t)) : lds;
    hipLaunchKernelGGL(rt_shade, dim3(blocks), dim3(RT_SHADE_THREADS), shade_lds, stream, *S, *F, level, setup_next);
    if (marks) (void)hipEventRecord(marks[3], stream);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_resolve(const DFrame* F, uint8_t* out, uint32_t bytes_per_pixel, uint32_t tiled,
                                        hipStream_t stream) {
    uint32_t blocks = (F->n_pixels + 255) / 256;
    hipLaunchKernelGGL(rt_resolve, dim3(blocks), dim3(256), 0, stream, *F, out, bytes_per_pixel, tiled);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_resolve_aux(const DFrame* F, uint32_t* prim, uint32_t* inst, uint8_t* flags, float* hit_t,
                                            hipStream_t stream) {
    uint32_t blocks = (F->n_pixels + 255) / 256;
    hipLaunchKernelGGL(rt_resolve_aux, dim3(blocks), dim3(256), 0, stream, *F, prim, inst, flags, hit_t);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_assemble(const uint8_t* gathered, uint8_t* frame, uint32_t width, uint32_t height,
                                         uint32_t tiles_x, uint32_t shard_count, uint32_t tiles_per_shard,
                                         uint32_t bytes_per_pixel, uint32_t n_views, hipStream_t stream) {
    const size_t n = (size_t)width * height * n_views;
    hipLaunchKernelGGL(rt_assemble, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, gathered, frame, width, height, tiles_x,
                       shard_count, tiles_per_shard, bytes_per_pixel, n_views);
    return hipGetLastError();
}
