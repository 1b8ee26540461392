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
//  * Rays of recursion depth d live in a queue in HBM (32 B per ray); one launch per depth consumes
//    queue d and appends the reflected / transmitted children to queue d+1.  A launch is a
//    persistent grid: every lane pulls rays from the queue until it is empty, so lanes whose ray is
//    finished are refilled and terminated rays are compacted away (wave-aggregated atomics on the
//    queue cursor, one per wave and refill).
//  * A lane is a small state machine: MAIN (closest hit of its ray) then one SHADOW state per light
//    (the shadow ray of raytracer.rs:78) - every trip round the loop traces exactly one ray through
//    the same inlined scene traversal, whatever state the lane is in.
//  * The BVH traversal stack of intersect.rs:240-256 is a per-lane column of LDS
//    (stack[entry][lane], conflict-free); an inner node is 64 B and carries the boxes of both
//    children; the triangles of a leaf are consecutive.
//
// Every arithmetic expression that feeds a hit decision keeps the operation order of the reference
// (see the citations at each function); the file must be compiled with -ffp-contract=off and
// without fast-math so that no FMA is formed and division / sqrt stay correctly rounded.
#include <hip/hip_runtime.h>

#include "rt_device.h"

#define RT_BLOCK_THREADS 256
#define RT_WAVE 64
#define RT_REFILL_CHUNK 64u
#define RT_STAGING_RECORDS 512u

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
__device__ __forceinline__ bool sphere_t(const DSphere& s, const Ray& r, float& t) {
    V3 c = ld3(s.c);
    float radius = s.r;
    V3 m = r.o - c;
    float a = dot3(r.d, r.d);
    float b = 2.0f * dot3(m, r.d);
    float cc = dot3(m, m) - radius * radius;
    V3 dn = normalize3(r.d);
    V3 proj = m - dn * dot3(m, dn);
    float disc = 4.0f * dot3(r.d, r.d) * (radius * radius - dot3(proj, proj));
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

// Diagnostic build (-DRT_STATS): per-launch totals of traversal events, see DCounters::dbg.
#ifdef RT_STATS
struct DbgStats { uint32_t inner, leaf, tris, setups, rounds, transitions; };
#define RT_STAT(x) (x)
#else
#define RT_STAT(x) ((void)0)
#endif
// Pruning slack, see trace_scene: relative part of every "is this box beyond the best hit" comparison.
#define RT_PRUNE_REL 1.015625f            // 1 + 2^-6

// Closest hit of `r` against the whole scene with the outcome of scene.rs:120-140.
//
// `stack` points at this lane's column of the wave's LDS stack: entry e of the reference column is
// stack[e * RT_WAVE], its box-entry distance is stack[(entries + e) * RT_WAVE].
//
// Mesh traversal visits fewer nodes than intersect.rs:228-269 does, without changing the result:
//  * A node's box is tested when its parent is visited (the parent holds both child boxes), with
//    the reference's arithmetic, so the set of boxes that "pass" is the reference's.
//  * Distance pruning: a box whose entry distance lies beyond the best candidate so far (or beyond
//    `limit_world`: the best hit of earlier primitives, or the light of a shadow ray) by more than
//    the slack cannot hold a candidate that compares lower.  The slack (RT_PRUNE_REL and the
//    instance's prune_abs) dominates the rounding error of Moeller-Trumbore's t for the triangle
//    sizes of a `prunable` mesh; other meshes are walked exactly as the reference walks them.
//  * Near child first.  The reference resolves equal t by visiting order (utils/mod.rs:17, strict
//    `<`); triangle slots are numbered in that visiting order, so "lower slot wins a tie" gives the
//    same winner in any visiting order.
//  * Rays with a non-finite component take the reference's order and no pruning; a NaN candidate
//    ends the walk because nothing can replace it.
// any_hit (shadow rays, raytracer.rs:171-187): the caller only needs `have && t < limit_world`
// (or `have` for a directional light), so the walk stops at the first candidate that settles it.
__device__ __forceinline__ void trace_scene(const DScene& S, const Ray& r, uint32_t* stack, Closest& best,
                                            const float limit_world, const bool any_hit, const bool directional
#ifdef RT_STATS
                                            , DbgStats& dbg
#endif
                                            ) {
    best.have = false;
    best.t = 0.0f;
    best.prim = RT_PRIM_NONE;
    best.inst = RT_INSTANCE_NONE;
    best.tl = best.u = best.v = 0.0f;
    best.slot = 0;
    const bool ray_finite = finite3(r.o) && finite3(r.d);
    // a shadow ray may stop as soon as one candidate proves "shadow"; with NaNs about only the
    // directional case is decided by the mere existence of a candidate
    const bool early = any_hit && (directional || ray_finite);

    // planes.as_slice().intersect(ray): intersect.rs:20-36 then the merge of scene.rs:129
    {
        bool g_have = false; float g_t = 0.0f; uint32_t g_i = 0;
        for (uint32_t i = 0; i < S.n_planes; ++i) {
            float t;
            if (plane_t(S.planes[i], r, t) && closer(g_have, g_t, t)) { g_have = true; g_t = t; g_i = i; }
        }
        if (g_have && closer(best.have, best.t, g_t)) { best.have = true; best.t = g_t; best.prim = ((uint32_t)RT_PRIM_PLANE << 30) | g_i; }
    }
    if (early && best.have && (directional || best.t < limit_world)) return;
    {
        bool g_have = false; float g_t = 0.0f; uint32_t g_i = 0;
        for (uint32_t i = 0; i < S.n_spheres; ++i) {
            float t;
            if (sphere_t(S.spheres[i], r, t) && closer(g_have, g_t, t)) { g_have = true; g_t = t; g_i = i; }
        }
        if (g_have && closer(best.have, best.t, g_t)) { best.have = true; best.t = g_t; best.prim = ((uint32_t)RT_PRIM_SPHERE << 30) | g_i; }
    }
    if (early && best.have && (directional || best.t < limit_world)) return;
    {
        bool g_have = false; float g_t = 0.0f; uint32_t g_i = 0;
        for (uint32_t i = 0; i < S.n_triangles; ++i) {
            const DTriangle& tr = S.triangles[i];
            float t, u, v;
            if (triangle_tuv(ld3(tr.p0), ld3(tr.e1), ld3(tr.e2), r, t, u, v) && closer(g_have, g_t, t)) {
                g_have = true; g_t = t; g_i = i;
            }
        }
        if (g_have && closer(best.have, best.t, g_t)) { best.have = true; best.t = g_t; best.prim = ((uint32_t)RT_PRIM_TRIANGLE << 30) | g_i; }
    }
    if (early && best.have && (directional || best.t < limit_world)) return;

    uint32_t* stack_t = stack + S.stack_entries * RT_WAVE;
    const float o_l1 = fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z);

    // Mesh instances in Vec order: intersect.rs:290-326 around 228-269.  Every lane walks its own
    // sequence instance 0, 1, ... and steps through one BVH node per trip of the loop below, so a lane
    // that is done with an instance does not wait for the lanes still inside it.  Entering and leaving
    // an instance (transforms, root test, world-space t) is the expensive, divergent part: lanes that
    // need it wait until a quarter of the wave does, or until no lane has a node left to visit.
    uint32_t next_inst = 0, cur_inst = 0;
    bool in_mesh = false, done = false;
    uint32_t sp = 0;
    Ray lr;
    lr.o = lr.d = mk(0, 0, 0);
    V3 inv = mk(0, 0, 0);
    bool fast = false;
    float limit = RT_INF, prune_abs = 0.0f, slack_w = 0.0f;
    bool m_have = false; float m_t = 0.0f, m_u = 0.0f, m_v = 0.0f; uint32_t m_slot = 0;

    for (;;) {
        const bool waiting = !done && sp == 0;
        const unsigned long long wait_mask = __ballot(waiting);
        const unsigned long long run_mask = __ballot(!done && sp > 0);
        if (!wait_mask && !run_mask) break;
        RT_STAT(dbg.rounds += (__ffsll((long long)(wait_mask | run_mask)) - 1 == (int)(threadIdx.x & 63)) ? 1u : 0u);
        if (waiting && (run_mask == 0 || __popcll(wait_mask) >= 16)) {
            RT_STAT(dbg.transitions += 1);
            if (in_mesh) {
                in_mesh = false;
                if (m_have) {
                    // world-space t of this instance's closest hit: intersect.rs:312-319
                    const DInstance& inst = S.instances[cur_inst];
                    float4 h2 = S.tri_hot[3 * m_slot + 2];
                    V3 fn = mk(h2.y, h2.z, h2.w);
                    V3 p = lr.o + lr.d * m_t;
                    bool front = dot3(fn, lr.d) < 0.0f;
                    V3 sn = front ? fn : -fn;
                    V3 pos_local = p + sn * 1e-4f;
                    V3 pos_world = transform34(inst.model, pos_local);
                    float tw = distance3(r.o, pos_world);
                    if (closer(best.have, best.t, tw)) {
                        best.have = true; best.t = tw; best.inst = cur_inst; best.slot = m_slot;
                        best.tl = m_t; best.u = m_u; best.v = m_v;
                        best.prim = ((uint32_t)RT_PRIM_MESH_TRIANGLE << 30) | __float_as_uint(S.tri_cold[3 * m_slot].w);
                    }
                    if (early && best.have && (directional || best.t < limit_world)) done = true;
                }
            }
            if (best.have && best.t != best.t) done = true;     // a NaN closest hit is final (utils/mod.rs:17)
            while (!done && !in_mesh && next_inst < S.n_instances) {
                const uint32_t ii = next_inst++;
                RT_STAT(dbg.setups += 1);
                const DInstance& inst = S.instances[ii];
                if (inst.root_ref == RT_REF_NONE) continue;     // empty node range, intersect.rs:235-237
                lr.o = transform34(inst.model_inverse, r.o);
                V3 dv = transform34(inst.rotation_scale_inverse, r.d);
                float dn = sqrtf(dot3(dv, dv));
                lr.d = mk(dv.x / dn, dv.y / dn, dv.z / dn);     // glm::normalize
                inv = mk(1.0f / lr.d.x, 1.0f / lr.d.y, 1.0f / lr.d.z);
                fast = inst.prunable && ray_finite && finite3(lr.o) && finite3(lr.d);
                slack_w = 2.0f * inst.world_guard;
                // mesh-local bound on useful t: a hit at world distance w lies at local t = w * dn (up to world_guard)
                limit = RT_INF;
                if (fast) {
                    float lw = limit_world;
                    if (best.have && best.t < lw) lw = best.t;
                    if (lw < RT_INF) limit = (lw + slack_w + 1.52587890625e-05f * (o_l1 + lw)) * dn * 1.0009765625f;
                }
                prune_abs = inst.prune_abs;
                float root_tmin;
                if (!slab(inst.root_lo[0], inst.root_lo[1], inst.root_lo[2], inst.root_hi[0], inst.root_hi[1], inst.root_hi[2], lr.o, inv, root_tmin))
                    continue;
                if (fast && root_tmin > limit * RT_PRUNE_REL + prune_abs) continue;
                m_have = false; m_t = 0.0f; m_u = 0.0f; m_v = 0.0f; m_slot = 0;
                stack[0] = inst.root_ref;
                stack_t[0] = __float_as_uint(root_tmin);
                sp = 1;
                cur_inst = ii;
                in_mesh = true;
            }
            if (!in_mesh) done = true;
        }
        if (!done && sp > 0) {
            --sp;
            const uint32_t ref = stack[sp * RT_WAVE];
            const float entry = __uint_as_float(stack_t[sp * RT_WAVE]);
            float cut = RT_INF;                                 // boxes entered beyond this cannot matter
            if (fast) {
                float lim = (m_have && m_t < limit) ? m_t : limit;
                cut = lim * RT_PRUNE_REL + prune_abs;
            }
            if (entry > cut) continue;
            if (ref & RT_REF_LEAF_BIT) {
                const uint32_t first = ref & RT_REF_SLOT_MASK;
                const uint32_t cnt = (ref >> RT_REF_COUNT_SHIFT) & 7u;
                bool improved = false;
                RT_STAT(dbg.leaf += 1);
                RT_STAT(dbg.tris += cnt);
                for (uint32_t k = 0; k < cnt; ++k) {
                    const uint32_t slot = first + k;
                    float4 h0 = S.tri_hot[3 * slot];
                    float4 h1 = S.tri_hot[3 * slot + 1];
                    float4 h2 = S.tri_hot[3 * slot + 2];
                    float t, u, v;
                    if (triangle_tuv(mk(h0.x, h0.y, h0.z), mk(h0.w, h1.x, h1.y), mk(h1.z, h1.w, h2.x), lr, t, u, v) &&
                        (!m_have || t < m_t || (t == m_t && slot < m_slot))) {
                        m_have = true; m_t = t; m_u = u; m_v = v; m_slot = slot;
                        improved = true;
                    }
                }
                if (m_have && m_t != m_t) sp = 0;               // NaN candidate: final
                if (early && improved) {
                    // would this candidate already prove "shadow"?  The instance's final hit has a local t
                    // that is not larger, hence a world t that is at most 2 * world_guard larger.
                    const DInstance& inst = S.instances[cur_inst];
                    float4 h2 = S.tri_hot[3 * m_slot + 2];
                    V3 fn = mk(h2.y, h2.z, h2.w);
                    V3 p = lr.o + lr.d * m_t;
                    V3 sn = (dot3(fn, lr.d) < 0.0f) ? fn : -fn;
                    float tw = distance3(r.o, transform34(inst.model, p + sn * 1e-4f));
                    if (directional || tw < limit_world - (slack_w + 6.103515625e-05f * (o_l1 + limit_world))) {
                        best.have = true; best.t = tw;
                        done = true;
                        sp = 0;
                    }
                }
            } else {
                RT_STAT(dbg.inner += 1);
                const float4 n0 = S.nodes[4 * ref], n1 = S.nodes[4 * ref + 1], n2 = S.nodes[4 * ref + 2], n3 = S.nodes[4 * ref + 3];
                float tl, tr;
                bool hl = slab(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, lr.o, inv, tl);
                bool hr = slab(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, lr.o, inv, tr);
                hl = hl && !(tl > cut);
                hr = hr && !(tr > cut);
                uint32_t rl = __float_as_uint(n3.x), rr = __float_as_uint(n3.y);
                // reference order: push left, push right, so that right is popped first (intersect.rs:253-256);
                // on the fast path the nearer child is popped first instead
                if (fast && tl < tr) {
                    uint32_t tu = rl; rl = rr; rr = tu;
                    float tf = tl; tl = tr; tr = tf;
                    bool tb = hl; hl = hr; hr = tb;
                }
                if (hl) { stack[sp * RT_WAVE] = rl; stack_t[sp * RT_WAVE] = __float_as_uint(tl); ++sp; }
                if (hr) { stack[sp * RT_WAVE] = rr; stack_t[sp * RT_WAVE] = __float_as_uint(tr); ++sp; }
            }
        }
    }
}

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
// poison bits per channel c: bit c = NaN, bit 3+c = +inf (or beyond the fixed-point range), bit 6+c = -inf
#define RT_FIXED_ONE 1099511627776.0     // 2^40
#define RT_FIXED_LIMIT 4194304.0f        // 2^22

__device__ __forceinline__ void add_channel(const DFrame& F, uint32_t pixel, int c, float v) {
    if (v == 0.0f) return;
#ifdef RT_ABLATE_ATOMICS
    if (v != 12345.678f) return;      // timing experiment only: results are wrong
#endif
    if (fabsf(v) < RT_FIXED_LIMIT) {
        long long q = __double2ll_rn((double)v * RT_FIXED_ONE);
        atomicAdd(reinterpret_cast<unsigned long long*>(F.acc + (size_t)c * F.n_pixels + pixel), (unsigned long long)q);
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
#define RT_META_PIXEL_MASK 0x01FFFFFFu
#define RT_META_KIND_SHIFT 25            // 0 = primary / reflected, 1 = transmitted
#define RT_META_DEPTH_SHIFT 26

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

// local pixel index -> canvas (col, row); false for padding pixels outside the frame
__device__ __forceinline__ bool pixel_coords(const DFrame& F, uint32_t p, uint32_t& col, uint32_t& row) {
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
__device__ __forceinline__ Ray primary_ray(const DFrame& F, uint32_t col, uint32_t row) {
    const float* m = F.screen_to_world;
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

enum { ST_IDLE = 0, ST_MAIN = 1, ST_SHADOW = 2, ST_EMPTY = 3 };

// Moves a wave's staged ray records to the output queue: one atomic reserves the space, the lanes copy.
// Records beyond the queue's capacity stay (moved to the front of the staging buffer) and `queue_full`
// is set.  Must be called by all 64 lanes.
__device__ __forceinline__ void flush_staging(DCounters* C, uint32_t level, uint32_t capacity, float4* q_out, float4* staging,
                                              uint32_t& stage_n, bool& queue_full, uint32_t lane) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&C->queue_count[level & 1], stage_n);
    base = __shfl(base, 0);
    uint32_t fit = 0;
    if (base < capacity) fit = (capacity - base < stage_n) ? capacity - base : stage_n;
    for (uint32_t i = lane; i < fit; i += RT_WAVE) {
        float4 a = staging[2 * (size_t)i], b = staging[2 * (size_t)i + 1];
        q_out[2 * ((size_t)base + i)] = a;
        q_out[2 * ((size_t)base + i) + 1] = b;
    }
    if (fit < stage_n) {
        queue_full = true;
        const uint32_t rest = stage_n - fit;
        if (fit > 0) {
            for (uint32_t i0 = 0; i0 < rest; i0 += RT_WAVE) {       // ascending: a pass never overwrites a later pass's source
                uint32_t i = i0 + lane;
                float4 a = make_float4(0, 0, 0, 0), b = a;
                if (i < rest) { a = staging[2 * (size_t)(fit + i)]; b = staging[2 * (size_t)(fit + i) + 1]; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (i < rest) { staging[2 * (size_t)i] = a; staging[2 * (size_t)i + 1] = b; }
            }
        }
        stage_n = rest;
    } else {
        stage_n = 0;
    }
}

}  // namespace

// One pass over the rays of one recursion depth.  `level` 0 generates primary rays from pixel
// indices; other levels read queue[(level + 1) & 1] and all append to queue[level & 1].
__global__ void __launch_bounds__(RT_BLOCK_THREADS)
rt_trace_level(const DScene S, const DFrame F, const uint32_t level) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t lane = threadIdx.x & (RT_WAVE - 1);
    const uint32_t wave = threadIdx.x / RT_WAVE;
    uint32_t* stack = lds_stack + wave * (2 * S.stack_entries * RT_WAVE) + lane;
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;

    DCounters* C = F.counters;
    const float4* q_in = F.queue[(level + 1) & 1];
    float4* q_out = F.queue[level & 1];
    uint32_t n_in;
    if (level == 0) {
        n_in = F.n_pixels;
    } else {
        n_in = C->queue_count[(level + 1) & 1];
        if (n_in > F.queue_capacity) n_in = F.queue_capacity;
    }
    float4* spill = F.spill + (size_t)gtid * F.max_depth * 2;
    uint32_t spill_n = 0;

    uint32_t n_primary = 0, n_shadow = 0, n_reflected = 0, n_transmitted = 0, n_spilled = 0;

    // Input: chunk c is the rays [c * RT_REFILL_CHUNK, (c + 1) * RT_REFILL_CHUNK); wave w of W takes the
    // chunks w, w + W, w + 2W, ... - neighbouring chunks go to different waves, and nobody touches a
    // shared cursor (a contended device-wide atomic costs about 10 ns, serialised over all waves).
    const uint32_t gwave = gtid / RT_WAVE;
    const uint32_t n_waves = gridDim.x * (RT_BLOCK_THREADS / RT_WAVE);
    uint32_t my_chunk = gwave;
    uint32_t chunk_next = 0, chunk_end = 0;     // wave-uniform: what is left of the current chunk
    bool exhausted = false;

    // Output: children are staged in a wave-private buffer and moved to the next depth's queue in
    // bulk, one atomic per move.  If that queue is full the wave keeps them and traces them itself.
    float4* staging = F.staging + (size_t)gwave * RT_STAGING_RECORDS * 2;
    uint32_t stage_n = 0;                       // wave-uniform
    bool queue_full = false;                    // wave-uniform

#ifdef RT_STATS
    DbgStats dbg = {0, 0, 0, 0, 0, 0};
    uint32_t dbg_iters = 0;
#endif
    int state = ST_IDLE;
    RayRec cur;                 // the lane's MAIN ray
    cur.ray.o = cur.ray.d = mk(0, 0, 0);
    cur.throughput = 0.0f;
    cur.meta = 0;
    // state carried from MAIN to the SHADOW states
    Hitpoint hit;
    hit.position = hit.hit_normal = hit.position_for_refraction = mk(0, 0, 0);
    hit.on_frontside = false;
    hit.material = 0;
    V3 color = mk(0, 0, 0);
    uint32_t light_i = 0;
    Ray sray;
    sray.o = sray.d = mk(0, 0, 0);

    for (;;) {
        // ---- refill idle lanes: own spilled children first, then the input queue ----------------
        if (state == ST_IDLE && spill_n > 0) {
            --spill_n;
            cur = load_rec(spill, spill_n);
            state = ST_MAIN;
        }
        {
            bool want = state == ST_IDLE;
            unsigned long long mask = __ballot(want);
            if (mask) {
                const uint32_t asking = __popcll(mask);
                const uint32_t r = __popcll(mask & ((1ull << lane) - 1ull));
                // staged children that can no longer go to the next depth's queue come first (newest first)
                const uint32_t from_stage = queue_full ? (asking < stage_n ? asking : stage_n) : 0u;
                if (want && r < from_stage) {
                    cur = load_rec(staging, stage_n - 1 - r);
                    state = ST_MAIN;
                    ++n_spilled;
                }
                stage_n -= from_stage;
                if (asking > from_stage) {
                    if (chunk_next == chunk_end && !exhausted) {
                        if ((unsigned long long)my_chunk * RT_REFILL_CHUNK >= n_in) {
                            exhausted = true;
                        } else {
                            chunk_next = my_chunk * RT_REFILL_CHUNK;
                            chunk_end = (n_in - chunk_next < RT_REFILL_CHUNK) ? n_in : chunk_next + RT_REFILL_CHUNK;
                            my_chunk += n_waves;
                        }
                    }
                    const uint32_t avail = chunk_end - chunk_next;
                    const uint32_t asking2 = asking - from_stage;
                    if (want && r >= from_stage) {
                        const uint32_t r2 = r - from_stage;
                        if (r2 < avail) {
                            uint32_t idx = chunk_next + r2;
                            if (level == 0) {
                                uint32_t col, row;
                                if (pixel_coords(F, idx, col, row)) {
                                    cur.ray = primary_ray(F, col, row);
                                    cur.throughput = 1.0f;
                                    cur.meta = idx;
                                    state = ST_MAIN;
                                }                               // padding pixel: stay idle, ask again
                            } else {
                                cur = load_rec(q_in, idx);
                                state = ST_MAIN;
                            }
                        } else if (exhausted && avail == 0) {
                            state = ST_EMPTY;
                        }
                    }
                    chunk_next += asking2 < avail ? asking2 : avail;
                }
            }
        }
        if (__all(state == ST_EMPTY)) {
            // nothing left to trace: hand the staged children over; whatever does not fit stays here
            if (stage_n > 0 && !queue_full) flush_staging(C, level, F.queue_capacity, q_out, staging, stage_n, queue_full, lane);
            if (stage_n == 0) break;
            state = ST_IDLE;
            continue;
        }
        RT_STAT(dbg_iters += (lane == 0) ? 1u : 0u);

        // ---- trace one ray per busy lane -------------------------------------------------------
        Closest cl;
        cl.have = false;
        {
            Ray tr = cur.ray;
            float limit_world = RT_INF;
            bool directional = false;
            if (state == ST_SHADOW) {
                tr = sray;
                const DLight& L = S.lights[light_i];
                directional = L.pos[3] == 0.0f;
                if (!directional) limit_world = distance3(sray.o, ld3(L.world));      // raytracer.rs:177-178
            }
            if (state == ST_MAIN || state == ST_SHADOW) trace_scene(S, tr, stack, cl, limit_world, state == ST_SHADOW, directional
#ifdef RT_STATS
                                                                    , dbg
#endif
                                                                    );
        }
        // children produced by this trip round the loop; queued together further down
        bool emit_r = false, emit_t = false;
        RayRec rec_r, rec_t;
        rec_r.ray.o = rec_r.ray.d = rec_t.ray.o = rec_t.ray.d = mk(0, 0, 0);
        rec_r.throughput = rec_t.throughput = 0.0f;
        rec_r.meta = rec_t.meta = 0;

        if (state == ST_MAIN) {
            const uint32_t pixel = cur.meta & RT_META_PIXEL_MASK;
            const uint32_t depth = cur.meta >> RT_META_DEPTH_SHIFT;
            const uint32_t kind = (cur.meta >> RT_META_KIND_SHIFT) & 1u;
            if (depth == 0) ++n_primary; else if (kind) ++n_transmitted; else ++n_reflected;
            if (!cl.have) {
                // raytrace returned None: the caller substitutes the background (main.rs:58-61, raytracer.rs:92,110,112)
                add_color(F, pixel, background(S, cur.ray.d) * cur.throughput);
                state = ST_IDLE;
            } else {
                hit = finalize_hit(S, cur.ray, cl);
                const RtMaterial& mat = S.materials[hit.material];
                const uint32_t mtype = mat.material_type;
                const bool returns_some = !(mtype == RT_MATERIAL_PHONG && S.n_lights == 0);
                if (F.aux_flags) {
                    if (depth == 0) {
                        F.aux_prim[pixel] = cl.prim;
                        F.aux_inst[pixel] = cl.inst;
                        atomicOr(F.aux_flags + pixel, RT_FLAG_HIT | (hit.on_frontside ? RT_FLAG_FRONTSIDE : 0u));
                    } else if (depth == 1 && returns_some) {
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
                    rec_r.meta = pixel | ((depth + 1) << RT_META_DEPTH_SHIFT);
                    rec_t.ray.o = hit.position_for_refraction; rec_t.ray.d = td;
                    rec_t.throughput = cur.throughput * k_transmitted;
                    rec_t.meta = pixel | (1u << RT_META_KIND_SHIFT) | ((depth + 1) << RT_META_DEPTH_SHIFT);
                    state = ST_IDLE;
                    if (depth + 1 < F.max_depth) {
                        emit_r = emit_t = true;
                    } else {
                        // raytrace_impl returns None without tracing (raytracer.rs:66): background of each child
                        add_color(F, pixel, background(S, rd) * rec_r.throughput);
                        add_color(F, pixel, background(S, td) * rec_t.throughput);
                    }
                } else if (S.n_lights > 0) {
                    // shade_phong, raytracer.rs:75-84: one shadow ray per light, in order
                    light_i = 0;
                    color = mk(0, 0, 0);
                    state = ST_SHADOW;
                } else if (mtype == RT_MATERIAL_PHONG) {
                    // shade() returned None: as a miss
                    add_color(F, pixel, background(S, cur.ray.d) * cur.throughput);
                    state = ST_IDLE;
                } else {
                    state = ST_SHADOW;       // ReflectAndPhong without lights: goes straight to the finish below
                    light_i = 0;
                    color = mk(0, 0, 0);
                }
            }
        } else if (state == ST_SHADOW) {
            // raytracer.rs:171-187 + radiance 137-153 for light `light_i`
            ++n_shadow;
            const DLight& L = S.lights[light_i];
            const RtMaterial& mat = S.materials[hit.material];
            bool directional = L.pos[3] == 0.0f;
            bool is_shadow = false;
            if (cl.have) {
                if (directional) is_shadow = true;
                else is_shadow = cl.t < distance3(sray.o, ld3(L.world));
            }
            if (F.aux_flags && is_shadow && light_i < 4 && (cur.meta >> RT_META_DEPTH_SHIFT) == 0)
                atomicOr(F.aux_flags + (cur.meta & RT_META_PIXEL_MASK), RT_FLAG_SHADOW_LIGHT0 << light_i);
            V3 l = sray.d;                                     // = get_hitpoint_to_light_unit_vector
            V3 n = hit.hit_normal;
            V3 v = -cur.ray.d;
            V3 rv = reflected_dir(l, n);
            float l_dot_n = fmaxf(dot3(l, n), 0.0f);
            float r_dot_v = fmaxf(dot3(rv, v), 0.0f);
            V3 emissive = ld3(mat.emissive);
            V3 ambient = cmul(ld3(L.ambient), ld3(mat.ambient));
            V3 zero = mk(0.0f, 0.0f, 0.0f);
            V3 diffuse = is_shadow ? zero : cmul(ld3(L.diffuse), ld3(mat.diffuse)) * l_dot_n;
            V3 specular = is_shadow ? zero : cmul(ld3(L.specular), ld3(mat.specular)) * powf(r_dot_v, mat.shininess);
            V3 rad = emissive + ambient + diffuse + specular;
            color = (light_i == 0) ? rad : color + rad;        // color::add_option
            ++light_i;
        }

        // ---- SHADOW bookkeeping: next light, or finish the shade ---------------------------------
        if (state == ST_SHADOW) {
            if (light_i < S.n_lights) {
                const DLight& L = S.lights[light_i];
                bool directional = L.pos[3] == 0.0f;
                V3 dir = directional ? mk(L.pos[0], L.pos[1], L.pos[2]) : ld3(L.world) - hit.position;
                sray.o = hit.position;
                sray.d = normalize3(dir);
            } else {
                const uint32_t pixel = cur.meta & RT_META_PIXEL_MASK;
                const uint32_t depth = cur.meta >> RT_META_DEPTH_SHIFT;
                const RtMaterial& mat = S.materials[hit.material];
                if (S.n_lights > 0) add_color(F, pixel, color * cur.throughput);
                state = ST_IDLE;
                if (mat.material_type == RT_MATERIAL_REFLECT_AND_PHONG) {
                    // raytracer.rs:86-94
                    V3 rd = normalize3(reflected_dir(-cur.ray.d, hit.hit_normal));
                    rec_r.ray.o = hit.position; rec_r.ray.d = rd;
                    rec_r.throughput = cur.throughput * 0.8f;  // REFLECTION_DIM_FACTOR
                    rec_r.meta = pixel | ((depth + 1) << RT_META_DEPTH_SHIFT);
                    if (depth + 1 < F.max_depth) emit_r = true;
                    else add_color(F, pixel, background(S, rd) * rec_r.throughput);
                }
            }
        }

        // ---- stage the children; move them to the next depth's queue when the buffer fills up ------
        {
            unsigned long long mask_r = __ballot(emit_r), mask_t = __ballot(emit_t);
            uint32_t total = __popcll(mask_r) + __popcll(mask_t);
            if (total) {
                unsigned long long lt = (1ull << lane) - 1ull;
                uint32_t at_r = stage_n + __popcll(mask_r & lt);
                uint32_t at_t = stage_n + __popcll(mask_r) + __popcll(mask_t & lt);
                // children that do not fit are kept by this lane (transmitted first, so that the
                // reflected one is traced first, as in the reference's evaluation order)
                if (emit_t) {
                    if (at_t < RT_STAGING_RECORDS) store_rec(staging, at_t, rec_t);
                    else { store_rec(spill, spill_n, rec_t); ++spill_n; ++n_spilled; }
                }
                if (emit_r) {
                    if (at_r < RT_STAGING_RECORDS) store_rec(staging, at_r, rec_r);
                    else { store_rec(spill, spill_n, rec_r); ++spill_n; ++n_spilled; }
                }
                stage_n = (stage_n + total < RT_STAGING_RECORDS) ? stage_n + total : RT_STAGING_RECORDS;
                if (!queue_full && stage_n > RT_STAGING_RECORDS - 2 * RT_WAVE)
                    flush_staging(C, level, F.queue_capacity, q_out, staging, stage_n, queue_full, lane);
            }
        }
    }

    // ---- ray statistics: every wave owns a row of F.wave_stats (no atomics) ------------------------
    {
        uint32_t vals[6] = {n_primary + n_shadow + n_reflected + n_transmitted, n_primary, n_shadow, n_reflected, n_transmitted, n_spilled};
        for (int k = 0; k < 6; ++k) {
            uint32_t v = vals[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v) F.wave_stats[(size_t)gwave * 8 + k] += v;
        }
    }
#ifdef RT_STATS
    {
        uint32_t dv[7] = {dbg.inner, dbg.leaf, dbg.tris, dbg.setups, dbg.rounds, dbg.transitions, dbg_iters};
        for (int k = 0; k < 7; ++k) {
            uint32_t v = dv[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v) atomicAdd(&C->dbg[level][k], (unsigned long long)v);
        }
    }
#endif
}

// Resets the count of the queue that the next pass will fill.
__global__ void rt_next_level(DCounters* C, uint32_t next_level) {
    C->queue_count[next_level & 1] = 0;
}

// color.rs:42-52 + canvas layout.  One thread per local pixel.
//  tiled == 0: row-major frame (bytes_per_pixel * width * height); padding pixels are skipped
//  tiled == 1: this shard's tiles back to back, pixels row-major inside a tile, padding written as 0
__global__ void rt_resolve(const DFrame F, uint8_t* out, uint32_t bytes_per_pixel, uint32_t tiled) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= F.n_pixels) return;
    uint32_t col = 0, row = 0;
    bool inside = pixel_coords(F, p, col, row);
    uint8_t rgb[3] = {0, 0, 0};
    if (inside) {
        unsigned int poison = F.poison[p];
        for (int c = 0; c < 3; ++c) {
            float v = (float)((double)F.acc[(size_t)c * F.n_pixels + p] * (1.0 / RT_FIXED_ONE));
            bool is_nan = (poison >> c) & 1u, pinf = (poison >> (3 + c)) & 1u, ninf = (poison >> (6 + c)) & 1u;
            if (is_nan || (pinf && ninf)) v = __uint_as_float(0x7FC00000u);
            else if (pinf) v = __uint_as_float(0x7F800000u);
            else if (ninf) v = __uint_as_float(0xFF800000u);
            float clamped = (v > 0.0f) ? ((v < 1.0f) ? v : 1.0f) : 0.0f;     // glm::clamp, NaN -> 0
            rgb[c] = (uint8_t)(int)(clamped * 255.0f);                        // `as u8` truncates
        }
    }
    size_t at;
    if (tiled) {
        uint32_t k = p / (RT_TILE_DIM * RT_TILE_DIM), q = p % (RT_TILE_DIM * RT_TILE_DIM);
        uint32_t b = q / 64, l = q % 64;
        uint32_t px = (b % 4) * 8 + (l % 8), py = (b / 4) * 8 + (l / 8);
        at = ((size_t)k * RT_TILE_DIM * RT_TILE_DIM + py * RT_TILE_DIM + px) * bytes_per_pixel;
    } else {
        if (!inside) return;
        at = ((size_t)row * F.width + col) * bytes_per_pixel;
    }
    out[at] = rgb[0]; out[at + 1] = rgb[1]; out[at + 2] = rgb[2];
    if (bytes_per_pixel == 4) out[at + 3] = inside ? 255 : 0;
}

// Scatters the aux planes (local pixel order) into row-major W*H arrays.
__global__ void rt_resolve_aux(const DFrame F, uint32_t* prim, uint32_t* inst, uint8_t* flags) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= F.n_pixels) return;
    uint32_t col, row;
    if (!pixel_coords(F, p, col, row)) return;
    size_t at = (size_t)row * F.width + col;
    unsigned int fl = F.aux_flags[p];
    bool hit = fl & RT_FLAG_HIT;
    if (prim) prim[at] = hit ? F.aux_prim[p] : RT_PRIM_NONE;
    if (inst) inst[at] = hit ? F.aux_inst[p] : RT_INSTANCE_NONE;
    if (flags) flags[at] = (uint8_t)fl;
}

// Gathered shard buffers (rank-major, tile-major inside a rank) -> row-major frame.
// One thread per frame pixel.
__global__ void rt_assemble(const uint8_t* gathered, uint8_t* frame, uint32_t width, uint32_t height, uint32_t tiles_x,
                            uint32_t shard_count, uint32_t tiles_per_shard, uint32_t bytes_per_pixel) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    uint32_t row = i / width, col = i % width;
    uint32_t tile = (row / RT_TILE_DIM) * tiles_x + col / RT_TILE_DIM;
    uint32_t shard = tile % shard_count, k = tile / shard_count;
    size_t src = (((size_t)shard * tiles_per_shard + k) * RT_TILE_DIM * RT_TILE_DIM +
                  (row % RT_TILE_DIM) * RT_TILE_DIM + col % RT_TILE_DIM) * bytes_per_pixel;
    size_t dst = (size_t)i * bytes_per_pixel;
    for (uint32_t c = 0; c < bytes_per_pixel; ++c) frame[dst + c] = gathered[src + c];
}

// ---- launch wrappers (called from rt_api.hip) ----------------------------------------------------

static size_t trace_lds_bytes(uint32_t stack_entries) {
    // two words per stack entry (child reference, box entry distance), one column per lane
    return (size_t)(RT_BLOCK_THREADS / RT_WAVE) * 2 * stack_entries * RT_WAVE * sizeof(uint32_t);
}

// Resident blocks of the trace kernel per compute unit (registers and the LDS stacks decide).
extern "C" int rt_query_trace_blocks_per_cu(uint32_t stack_entries) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, rt_trace_level, RT_BLOCK_THREADS, trace_lds_bytes(stack_entries)) != hipSuccess)
        return -1;
    return n;
}

extern "C" hipError_t rt_launch_trace_level(const DScene* S, const DFrame* F, uint32_t level, uint32_t blocks,
                                            hipStream_t stream) {
    size_t lds = trace_lds_bytes(S->stack_entries);
    hipLaunchKernelGGL(rt_trace_level, dim3(blocks), dim3(RT_BLOCK_THREADS), lds, stream, *S, *F, level);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_next_level(DCounters* C, uint32_t next_level, hipStream_t stream) {
    hipLaunchKernelGGL(rt_next_level, dim3(1), dim3(1), 0, stream, C, next_level);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_resolve(const DFrame* F, uint8_t* out, uint32_t bytes_per_pixel, uint32_t tiled,
                                        hipStream_t stream) {
    uint32_t blocks = (F->n_pixels + 255) / 256;
    hipLaunchKernelGGL(rt_resolve, dim3(blocks), dim3(256), 0, stream, *F, out, bytes_per_pixel, tiled);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_resolve_aux(const DFrame* F, uint32_t* prim, uint32_t* inst, uint8_t* flags,
                                            hipStream_t stream) {
    uint32_t blocks = (F->n_pixels + 255) / 256;
    hipLaunchKernelGGL(rt_resolve_aux, dim3(blocks), dim3(256), 0, stream, *F, prim, inst, flags);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_assemble(const uint8_t* gathered, uint8_t* frame, uint32_t width, uint32_t height,
                                         uint32_t tiles_x, uint32_t shard_count, uint32_t tiles_per_shard,
                                         uint32_t bytes_per_pixel, hipStream_t stream) {
    uint32_t n = width * height;
    hipLaunchKernelGGL(rt_assemble, dim3((n + 255) / 256), dim3(256), 0, stream, gathered, frame, width, height, tiles_x,
                       shard_count, tiles_per_shard, bytes_per_pixel);
    return hipGetLastError();
}
