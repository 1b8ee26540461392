// Device-side data layout of the gfx950 renderer, shared by the host upload code and the kernels.
//
// Everything the per-ray code reads is laid out for the way a wave64 reads it:
//  * the few analytic primitives, lights and materials are small arrays read with wave-uniform
//    indices (the compiler turns those into scalar loads);
//  * an inner BVH node is 64 B and holds the boxes of both children (one line, one round trip);
//  * mesh triangles are re-ordered so that the triangles of a leaf are consecutive, 48 B of "hot"
//    data per triangle (first vertex and the two edge vectors the Moeller-Trumbore test starts
//    from) and 48 B of "cold" data (normals) read only for the winning hit.
#pragma once
#include <stdint.h>

#include "../../include/rt_hip.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__
#else
#define RT_HD
struct float4 { float x, y, z, w; };
#endif

#define RT_STACK_LDS_MAX 12         // traversal stack entries per lane kept in LDS (four blocks of the BVH walk per compute unit)
#define RT_MAX_DEPTH_LIMIT 63      // depth is carried in 6 bits of a ray record
#define RT_MAX_PIXELS_PER_SHARD (1u << 28)
#define RT_META_RAY_LIMIT 0x7FFFFFFFull   // ray slots are 31-bit numbers

struct DLight {
    float pos[4];          // RtLight::position
    float world[3];        // position.xyz / position.w (raytracer.rs:162,177,215), unused when w == 0
    float ambient[3];
    float diffuse[3];
    float specular[3];
};

struct DPlane { float n[3]; float d; uint32_t material; };
struct DSphere { float c[3]; float r; uint32_t material; };

// Scene::triangles (free triangles): vertex 0, edges, normals
struct DTriangle {
    float p0[3], e1[3], e2[3];
    float n0[3], n1[3], n2[3];
    float fn[3];
    uint32_t material;
};

// 3x4 row-major top rows of a column-major Mat4: r[i][j] = M(i, j)
struct DMat34 { float r[3][4]; };

struct DInstance {
    DMat34 model_inverse;
    DMat34 rotation_scale_inverse;
    DMat34 model;
    DMat34 rotation_scale;
    float root_lo[3], root_hi[3];   // AABB of the mesh's root node
    uint32_t root_ref;              // child reference of the root (see DScene::nodes), RT_REF_NONE for an empty mesh
    uint32_t root_ref4;             // the same for the 4-wide tree (DScene::nodes4): what the BVH walk starts from
    int32_t material_override;      // -1 = none
    uint32_t prunable;              // 1: distance pruning and near-first ordering are provably result-preserving
    float prune_abs;                // absolute slack (mesh-local units) added to every pruning comparison
    float world_guard;              // bound on |t_world - k * t_local| for hits on this instance (world units)
    // World-space box around everything a ray can hit on this instance (the model transform of the root box,
    // padded).  A ray that misses it cannot hit the instance, so no local ray has to be derived for it.
    uint32_t cullable;              // 0: the box is not trusted (degenerate or inconsistent matrices)
    float cull_origin_limit;        // the box is only used for rays whose origin has a smaller L1 norm
    float world_lo[3], world_hi[3];
    // A ray that starts this far out (mesh-local units) on an axis resolves none of the mesh's boxes on that axis any more -
    // the slab test's products there all round to the same few values; with two such axes it would walk a good part of the
    // tree, and the pair is tested against every leaf by a whole block instead (traverse_heavy_pairs).  +inf where that is
    // not available.
    float heavy_origin;
    uint32_t leaf_begin, leaf_count;   // the mesh's leaves in DScene::leaf_boxes
};

// A BVH work item as the walk consumes it (64 B): the mesh-local ray and everything else it needs, so
// that starting a walk is one fetch.
//   q0 = (local origin, entry distance into the root box)     q1 = (local direction, local bound on useful t)
//   q2 = (1 / |rotation_scale_inverse * d|, shadow threshold, prune_abs, bits root reference)
//   q3 = (bits ray slot, bits instance | flags, 0, 0)
// shadow threshold: a candidate with local t * q2.x below it proves a shadow ray blocked (-inf for other rays).
#define RT_FAT_FAST 0x80000000u          // the walk may prune by distance and share subtrees (DInstance::prunable)
#define RT_FAT_DONE 0x40000000u          // no walk: the pair's result is already in item_results (ray traced on the exact path)
#define RT_FAT_HEAVY 0x20000000u         // not walked lane by lane: the pair is on the depth's list for traverse_heavy_pairs
#define RT_FAT_INSTANCE 0x1FFFFFFFu
#define RT_HEAVY_CAP 1024u               // pairs per depth traverse_heavy_pairs takes; further ones are walked as usual
#define RT_MAX_INSTANCES (1u << 20)      // mesh instances of a scene (rt_setup packs the index into 20 bits)

// A child reference: inner node index, or a leaf given by its first triangle slot and its count.
#define RT_REF_NONE 0xFFFFFFFFu
#define RT_REF_LEAF_BIT 0x80000000u
#define RT_REF_COUNT_SHIFT 28
#define RT_REF_SLOT_MASK 0x0FFFFFFFu

// Arrays in device memory.
struct DScene {
    RtBackground background;
    uint32_t n_lights, n_planes, n_spheres, n_triangles, n_instances;
    uint32_t n_tri_slots;        // entries of tri_hot / tri_cold
    uint32_t stack_entries;      // per-lane traversal stack entries a walk of any mesh's 4-wide tree can need
    uint32_t stack_lds;          // how many of them live in LDS (the rest, rarely reached, in DFrame::spill)
    uint32_t all_cullable;       // every instance has a usable culling box (DInstance::cullable)
    // Scenes with many instances: a binary tree over index ranges of the instances (node 0 = all of them, children
    // split a range in the middle), each node with the union of its instances' culling boxes, stored in pre-order: the
    // left child of node k is node k + 1, and everything up to `end` belongs to k's subtree.  Walking the array forwards
    // and jumping to `end` where a box is missed meets the instances in Vec order, which is the order the pairs of a ray
    // must be emitted in.
    //   node k: range_nodes[2k] = (box min, bits leaf word), [2k+1] = (box max, bits end)
    //   leaf word: 0 for an inner node, 0x80000000 | instance for a leaf, 0xFFFFFFFF for the leaf of an empty mesh
    //   range_limits[k] = smallest cull_origin_limit in the range
    uint32_t n_range_nodes;      // 0: rt_setup scans the instances linearly
    uint32_t range_depth;        // (unused: the walk of the tree keeps no stack)
    const float4* range_nodes;
    const float* range_limits;
    const DLight* lights;
    const RtMaterial* materials;
    const DPlane* planes;
    const DSphere* spheres;
    const DTriangle* triangles;
    const DInstance* instances;
    // Inner node i is 64 B and holds the boxes of its two children, so one fetch decides both:
    //   nodes[4*i]   = (L.min.x, L.min.y, L.min.z, L.max.x)   nodes[4*i+1] = (L.max.y, L.max.z, R.min.x, R.min.y)
    //   nodes[4*i+2] = (R.min.z, R.max.x, R.max.y, R.max.z)   nodes[4*i+3] = (bits ref L, bits ref R, 0, 0)
    const float4* nodes;
    // The same trees with every other level folded away, for the BVH walk: node i is 128 B, two to four entries in the
    // order the reference's walk would reach them, unused entries have the reference RT_REF_NONE.
    //   entry 0: nodes4[8*i]   = (min.xyz, max.x), nodes4[8*i+1].xy = max.yz      entry 1: nodes4[8*i+1].zw = min.xy, nodes4[8*i+2] = (min.z, max.xyz)
    //   entries 2, 3: the same in nodes4[8*i+3 .. 8*i+5]                          nodes4[8*i+6] = bits of the four references
    const float4* nodes4;
    // slot s: tri_hot[3*s] = (p0.xyz, e1.x), [3*s+1] = (e1.y, e1.z, e2.x, e2.y), [3*s+2] = (e2.z, fn.xyz)
    const float4* tri_hot;
    // slot s: tri_cold[3*s] = (n0.xyz, bits mesh_triangle_index), [3*s+1] = (n1.xyz, bits material), [3*s+2] = (n2.xyz, 0)
    const float4* tri_cold;
    // The leaves of every mesh whose tree nests, mesh after mesh (DInstance::leaf_begin): leaf k is leaf_boxes[2*k] =
    // (box min, box max.x), leaf_boxes[2*k+1] = (box max.yz, bits of the leaf's reference, 0).  In a tree whose boxes nest
    // a ray passes a leaf's box only if it passes the boxes of all the nodes above it (the slab test is monotone in the
    // box), so "the leaves whose own box the ray passes" are exactly the leaves the reference's walk tests.
    uint32_t n_leaf_boxes;
    const float4* leaf_boxes;
    // Per (material, light), at [material * n_lights + light]: what radiance() (raytracer.rs:137-153) can do with a shadow
    // verdict, RT_MOOT_* - see shadow_is_moot in rt_kernels.hip.  All zero: every shadow ray is traced.
    const uint32_t* moot_flags;
};

#define RT_MOOT_DIFFUSE_ZERO 1u       // light.diffuse * material.diffuse is +-0 in every channel
#define RT_MOOT_DIFFUSE_FINITE 2u     // ... is finite in every channel
#define RT_MOOT_SPECULAR_ZERO 4u      // light.specular * material.specular is +-0 in every channel and shininess >= 0
#define RT_MOOT_SPECULAR_FINITE 8u    // ... is finite in every channel and shininess > 0

// The cameras of the views of a frame (rt_generate's own argument): column-major screen_to_world matrices.
// (as many as one launch of rt_generate takes as its argument; a pass of more views is generated by several launches)
#define RT_VIEWS_PER_LAUNCH 32
struct DViews { float m[RT_VIEWS_PER_LAUNCH][16]; };

// Rows of DFrame::counts at depth d.
#define RT_ROW_RAYS 0    // the rays traced at depth d (written by rt_generate, or by rt_shade at depth d - 1)
#define RT_ROW_HITS 1    // the hit records of depth d - 1, whose shadow rays are traced at depth d
#define RT_ROW_FAT 2     // the walk items of depth d
#define RT_ROW_SLOW 3    // the rays of depth d left to rt_setup_slow
#define RT_COUNT_ROWS 4

// Per-frame device state.
//
// Work lists (rays, BVH work items, hit records) are cut into one segment per producer block plus
// one shared overflow segment.  A block appends to its own segment through a counter in LDS - no
// device-wide atomic is involved - and stores the final count at the end of the kernel; the
// consumer kernel turns the counts into a prefix table (in LDS) and walks the concatenation of
// all segments in interleaved 64-element chunks, so that work is spread evenly again whatever
// block produced it.
struct DFrame {
    uint32_t width, height;
    uint32_t tiles_x, tiles_y;
    uint32_t shard_index, shard_count;
    uint32_t n_local_tiles;          // tiles this shard owns (rounded up so every shard has the same number)
    uint32_t view_pixels;            // n_local_tiles * RT_TILE_DIM^2: local pixels of one view
    uint32_t n_views;                // views rendered together; local pixel p belongs to view p / view_pixels
    uint32_t n_pixels;               // n_views * view_pixels
    uint32_t max_depth;
    uint32_t mode;                   // RT_MODE_*
    uint32_t n_seg;                  // producer blocks + 1 (the last segment is the shared overflow segment)
    uint32_t seg_rays, ovf_rays;     // capacity of a block's ray segment / of the overflow segment
    uint32_t seg_fat, ovf_fat;       // (ray, instance) pairs that pass the exact root-box test: the walk items
    uint32_t n_fat_slots;            // entries of fat_items / item_results
    uint32_t seg_hits, ovf_hits;
    long long* acc;                  // [3][n_pixels] fixed-point colour sums, 2^-40 units
    double* acc_big;                 // [3][n_pixels] sums of the rare contributions beyond 2^22
    unsigned int* poison;            // [n_pixels] non-finite contributions, see rt_kernels.hip
    unsigned int* aux_flags;         // [n_pixels] or null
    unsigned int* aux_prim;          // [n_pixels] or null
    unsigned int* aux_inst;          // [n_pixels] or null
    float* aux_t;                    // [n_pixels] or null: t of the primary hit
    // Ray records of the current / next depth, 2 float4 each: the segmented list of the primary / reflected /
    // transmitted rays, slots below shadow_base.  The shadow ray of hit record h towards light l has the fixed slot number
    // shadow_base + h * n_lights + l (raytracer.rs:77-79: one per light per Phong-shaded hit) - a number for its candidate
    // key in `best` and for the walk items, not a record: the ray itself is rebuilt from the hit record wherever it is
    // needed (shadow_rec in rt_kernels.hip), so the shadow rays of a depth cost no ray traffic at all.
    float4* rays[2];
    uint32_t shadow_base;
    // Segment counts of the lists, one row of n_seg words per list and recursion depth, all zero when a frame starts (so
    // no kernel ever has to reset one): the row of list k (RT_ROW_*) at depth d starts at counts[(4 * d + k) * n_seg].
    // The data arrays behind them alternate between two copies with the depth's parity.
    unsigned int* counts;
    unsigned long long* best[2];     // [ray slots] closest candidate so far: t bits << 32 | order, see rt_kernels.hip
    float4* fat_items[2];            // walk items, 4 float4 per entry, segmented like the other lists
    float4* item_results[2];         // per walk item: local t, u, v, bits triangle slot
    unsigned int* slow_rays;         // ray slots rt_setup left to rt_setup_slow, segmented like the ray lists
    unsigned int* slow_flags;        // [RT_MAX_DEPTH_LIMIT + 2] set when a depth has such rays
    // rt_shade setting up the next depth itself: the slots a block took from the shared overflow segments, which are not in
    // its own segments - per block extra_rays_cap slots of rays, then extra_hits_cap of hit records
    uint32_t* extras;
    uint32_t extra_rays_cap, extra_hits_cap;
    uint2* spill;                    // traversal stack entries beyond DScene::stack_lds: entry e of walking lane g at [e * spill_lanes + g]
    uint32_t spill_lanes;            // lanes of the BVH walk's grid
    uint32_t walk_contiguous;        // the walk hands out runs of neighbouring pairs instead of interleaving them: 0 never, 1 always, 2 in launches with several chunks per wave
    uint32_t walk_packet_levels;     // the BVH walks of the depths below this number run as packet walks (rt_traverse_packet)
    uint32_t fuse_setup;             // rt_shade sets up the rays it makes (rt_setup only runs at depth 0)
    unsigned int* chunk_counter;     // [RT_MAX_DEPTH_LIMIT + 2] input chunks of the BVH walk handed out so far, per depth
    unsigned int* heavy_count;       // [RT_MAX_DEPTH_LIMIT + 2] pairs on the depth's list for traverse_heavy_pairs (may exceed RT_HEAVY_CAP: the rest are walked)
    unsigned int* heavy_items;       // [RT_MAX_DEPTH_LIMIT + 2][RT_HEAVY_CAP] their walk items
    float4* hits[2];                 // Phong-shaded hits waiting for their shadow rays, 3 float4 each
    unsigned int* wave_stats;        // [waves][8]: rays, primary, shadow, reflected, transmitted (rt_shade); 4-wide nodes visited, triangles tested (rt_traverse); pairs taken (traverse_heavy_pairs)
    // A list did not fit (the frame is then invalid): the kernels store the frame's sequence number here.  The host
    // only ever reads the word, so no report can be lost to a clear; frames of one handle run one after the other,
    // so the word names the latest frame that overflowed.
    unsigned int* overflow;
    uint32_t overflow_seq;           // this frame's sequence number on its handle, never 0
    // rt_setup_slow is only launched when slow_launched is set; a frame that turns out to hold rays for it while it is
    // not stores its sequence number to *needs_slow (host-mapped, like *overflow) and is invalid.
    unsigned int* needs_slow;
    uint32_t slow_launched;
};
