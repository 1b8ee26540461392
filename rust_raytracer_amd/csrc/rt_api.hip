// Host side of the C ABI declared in include/rt_hip.h: scene validation and upload, per-frame
// buffers, kernel sequencing.  No rendering arithmetic happens here - if the device code cannot
// run, every entry point fails with RT_ERR_DEVICE / RT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "rt_device.h"

extern "C" {
hipError_t rt_launch_generate(const DFrame* F, const float* matrices, uint32_t blocks, hipStream_t stream);
int rt_query_setup_blocks_per_cu(const DScene* S, uint32_t n_seg);
hipError_t rt_launch_level(const DScene* S, const DFrame* F, uint32_t level, uint32_t last, uint32_t blocks, uint32_t setup_blocks, uint32_t traverse_blocks,
                           hipStream_t stream, hipEvent_t* marks);      // launches rt_setup_slow iff F->slow_launched
hipError_t rt_launch_resolve(const DFrame* F, uint8_t* out, uint32_t bytes_per_pixel, uint32_t tiled, hipStream_t stream);
hipError_t rt_launch_resolve_aux(const DFrame* F, uint32_t* prim, uint32_t* inst, uint8_t* flags, float* hit_t, hipStream_t stream);
hipError_t rt_launch_assemble(const uint8_t* gathered, uint8_t* frame, uint32_t width, uint32_t height, uint32_t tiles_x,
                              uint32_t shard_count, uint32_t tiles_per_shard, uint32_t bytes_per_pixel, uint32_t n_views,
                              hipStream_t stream);
int rt_query_traverse_blocks_per_cu(uint32_t n_seg, uint32_t stack_entries);
}

static thread_local char g_error[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static int scratch_for_device(int device, void** out);

// instance counts above this get a range tree over the culling boxes (rt_setup walks it instead of scanning)
#define RT_RANGE_TREE_MIN 16
// the lists of a handle are doubled at most this often (ensure_frame refuses more than 2^30 rays per depth long before)
#define RT_MAX_CAPACITY_SHIFT 24

struct RtScene {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_trace_start = nullptr, ev_trace_stop = nullptr;
    bool timed = false;
    hipStream_t last_stream = nullptr;
    int n_cus = 0;
    uint32_t list_blocks = 0;           // grid of the list-producing kernels (= list segments per list)
    uint32_t setup_blocks = 0;          // grid of rt_setup where it is a kernel of its own: what is resident at once
    uint32_t fuse_setup = 1;            // DFrame::fuse_setup (RT_FUSE_SETUP overrides the choice made at creation)
    bool last_fused = false;            // ... as the frame enqueued last had it
    uint32_t traverse_blocks = 0;       // grid of the BVH walk
    // Host-mapped word into which the kernels store the sequence number of a frame whose lists did not fit.  The host
    // never writes it: what it has seen so far is `overflow_seen`, what it has not yet told the caller `overflow_unreported`.
    volatile unsigned int* h_overflow = nullptr;
    unsigned int* d_overflow = nullptr;
    uint32_t frame_seq = 0;             // sequence number of the last frame enqueued on this handle
    uint32_t overflow_seen = 0;
    bool overflow_unreported = false;
    // Second word of the same mapping: the sequence number of a frame that held rays for the exact path (rt_setup_slow)
    // while that kernel was not being launched.  Such a frame is invalid like one whose lists overflowed; from then on the
    // handle launches the kernel at every depth.
    uint32_t slow_seen = 0;
    bool slow_enabled = false;          // rt_setup_slow is launched at every depth
    uint32_t capacity_shift = 0;        // lists are (what a frame asks for << capacity_shift) entries long
    bool must_grow = false;             // a frame overflowed: the next one needs lists longer than it had
    // What the frames of this handle have needed, in entries per pixel of the fullest depth, for the three kinds of list
    // (rays, walk items, hit records) - taken whenever the caller has the lists counted (measure_lists).  Once known,
    // lists are made half as long again as that instead of the first guess of four rays per pixel.
    double seen_per_pixel[3] = {0.0, 0.0, 0.0};
    bool seen_valid = false;
    uint32_t seen_pixels = 0;           // pixel count of the frame measured last
    bool trim_pending = false;          // a measurement has come in since the buffers were made
    uint64_t last_used[3] = {0, 0, 0};  // entries in the fullest depth of the frame measured last
    size_t frame_bytes = 0;             // device memory held by the per-frame buffers
    uint32_t last_rows = 0;             // depths of the frame enqueued last whose count rows are valid
    bool last_noalloc = false;
    uint32_t noalloc_seq_lo = 1, noalloc_seq_hi = 0;     // sequence numbers of the latest run of rt_render_device_noalloc frames
    uint32_t walk_contiguous = 0;       // see DFrame::walk_contiguous
    uint32_t walk_packet_levels = 0;    // see DFrame::walk_packet_levels (RT_WALK_PACKET_LEVELS)
    void* scratch = nullptr;            // this device's copy of the static scratch array (rt_render_device_noalloc)
    uint32_t regrown = 0;               // times rt_render had to enlarge the lists for its last frame
    std::vector<void*> scene_allocs;
    DScene d{};
    bool camera_set = false;
    std::vector<float> views = std::vector<float>(16, 0.0f);      // screen_to_world of every view, 16 floats each
    uint32_t n_views = 1;
    uint32_t width = 0, height = 0;
    // per-frame buffers, reallocated when the key below changes
    struct FrameKey { uint32_t w = 0, h = 0, shard_count = 0, max_depth = 0, aux = 0, blocks = 0, debug = 0; uint64_t rays = 0, fat = 0, hits = 0; } key;
    DFrame f{};
    DFrame last_f{};                    // the frame enqueued last (its statistics are what rt_last_ray_counts reads)
    size_t last_stats_words = 0;
    std::vector<void*> frame_allocs;
    uint8_t* d_pixels = nullptr;        // staging for rt_render
    size_t d_pixels_bytes = 0;
    uint32_t *d_aux_prim_out = nullptr, *d_aux_inst_out = nullptr;
    uint8_t* d_aux_flags_out = nullptr;
    float* d_aux_t_out = nullptr;
    size_t aux_out_pixels = 0;
    uint32_t debug_queue_capacity = 0;
    size_t stats_words = 0;
    size_t pixel_clear_bytes = 0, counter_words = 0;
    // kernel timing mode (rt_set_kernel_timing): RT_N_LEVEL_KERNELS + 1 events per recursion depth
    bool kernel_timing = false;
    std::vector<hipEvent_t> marks;
    uint32_t marked_levels = 0;         // depths of the last frame that carry marks
};

template <class T>
static int upload(RtScene* s, const std::vector<T>& host, const T** out) {
    *out = nullptr;
    if (host.empty()) return RT_OK;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, host.size() * sizeof(T)));
    s->scene_allocs.push_back(p);
    HIP_TRY(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(p);
    return RT_OK;
}

static void mat34(const float* colmajor, DMat34* out) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) out->r[i][j] = colmajor[4 * j + i];
}

static float4 f4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
static float bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

namespace {
// One mesh's BVH re-laid out for the device (see DScene in rt_device.h).
struct MeshLayout {
    uint32_t root_ref = RT_REF_NONE;
    float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
    uint32_t stack_need = 1;         // entries the reference's own walk can need (intersect.rs:240-256)
    uint32_t root_ref4 = RT_REF_NONE;   // the same tree with every other level folded away (DScene::nodes4)
    uint32_t stack_need4 = 1;        // entries a depth-first walk of that tree can need
    float max_edge_product = 0.0f;   // max |e1| * |e2| over the mesh's triangles
    float max_edge_sum = 0.0f;       // max |e1| + |e2|
    // Every child box lies inside its parent's box and every triangle inside its leaf's box, as in a tree the
    // reference's builder makes (each node's box is the hull of its triangles, bvh/builder.rs:100-218).  The distance
    // pruning of the BVH walk, the culling boxes of rt_setup and the folding of levels all lean on it; a mesh whose
    // caller-supplied tree breaks it (refitted, loosened, hand-made) is walked without any of them.
    bool nested = true;
    uint32_t leaf_begin = 0, leaf_count = 0;     // the mesh's leaves in DScene::leaf_boxes (reference visiting order)
};
}  // namespace

// Walks the mesh's tree in the order the reference's traversal visits it (pop, push left, push right:
// intersect.rs:245-256, i.e. right-first pre-order).  Inner nodes are numbered in that order, the
// triangles of each leaf get consecutive slots in that order - so a lower slot means "visited earlier
// by the reference", which is what decides ties there (utils/mod.rs:17).
static int layout_mesh(const RtSceneDesc* desc, uint32_t mesh_index, std::vector<float4>& nodes, std::vector<float4>& hot,
                       std::vector<float4>& cold, std::vector<float4>& leaf_boxes, MeshLayout* out) {
    const RtMesh& m = desc->meshes[mesh_index];
    if (m.node_begin > m.node_end || m.node_end > desc->n_mesh_bvh_nodes)
        return fail(RT_ERR_OUT_OF_RANGE, "mesh %u: node range [%u, %u) outside the %u nodes", mesh_index, m.node_begin,
                    m.node_end, desc->n_mesh_bvh_nodes);
    if (m.node_begin == m.node_end) return RT_OK;            // empty BVH: never hit (intersect.rs:235-237)
    out->leaf_begin = (uint32_t)(leaf_boxes.size() / 2);
    struct Item { uint32_t src; uint32_t parent; bool right; uint32_t depth; };
    std::vector<Item> stack;
    stack.push_back({m.node_begin, 0xFFFFFFFFu, false, 0});
    size_t visited = 0;
    uint32_t need = 1;
    const size_t budget = 4 * (size_t)(m.node_end - m.node_begin) + 16;
    auto check_node = [&](uint32_t i) { return i < desc->n_mesh_bvh_nodes; };
    auto inside = [](const float* lo, const float* hi, const float* outer_lo, const float* outer_hi) {
        for (int c = 0; c < 3; ++c)
            if (!(lo[c] >= outer_lo[c] && hi[c] <= outer_hi[c])) return false;          // false for NaN as well
        return true;
    };
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        if (!check_node(it.src))
            return fail(RT_ERR_OUT_OF_RANGE, "mesh %u: BVH node index %u outside the %u nodes", mesh_index, it.src,
                        desc->n_mesh_bvh_nodes);
        if (++visited > budget) return fail(RT_ERR_INVALID_ARGUMENT, "mesh %u: BVH is not a tree", mesh_index);
        const RtBvhNode& n = desc->mesh_bvh_nodes[it.src];
        uint32_t ref;
        if (n.is_leaf) {
            if (n.triangle_count > RT_LEAF_TRIANGLES)
                return fail(RT_ERR_INVALID_ARGUMENT, "mesh %u: leaf with %u triangles", mesh_index, n.triangle_count);
            size_t first = hot.size() / 3;
            if (first + n.triangle_count > RT_REF_SLOT_MASK)
                return fail(RT_ERR_INVALID_ARGUMENT, "more than %u mesh triangles", RT_REF_SLOT_MASK);
            for (uint32_t k = 0; k < n.triangle_count; ++k) {
                uint32_t ti = n.triangle_indices[k];
                if (ti >= desc->n_mesh_triangles)
                    return fail(RT_ERR_OUT_OF_RANGE, "mesh %u: triangle index %u outside the %u mesh triangles", mesh_index, ti,
                                desc->n_mesh_triangles);
                const RtTriangle& t = desc->mesh_triangles[ti];
                if (t.material >= desc->n_materials)
                    return fail(RT_ERR_OUT_OF_RANGE, "mesh triangle %u: material %u outside the %u materials", ti, t.material,
                                desc->n_materials);
                for (int k3 = 0; k3 < 3; ++k3)
                    if (!inside(t.vertices[k3], t.vertices[k3], n.aabb_min, n.aabb_max)) out->nested = false;
                const float* p0 = t.vertices[0];
                // e1 = p1 - p0, e2 = p2 - p0: the first two operations of Triangle::intersect (intersect.rs:147-148)
                float e1[3], e2[3];
                for (int c = 0; c < 3; ++c) { e1[c] = t.vertices[1][c] - p0[c]; e2[c] = t.vertices[2][c] - p0[c]; }
                hot.push_back(f4(p0[0], p0[1], p0[2], e1[0]));
                hot.push_back(f4(e1[1], e1[2], e2[0], e2[1]));
                hot.push_back(f4(e2[2], t.face_normal[0], t.face_normal[1], t.face_normal[2]));
                cold.push_back(f4(t.normals[0][0], t.normals[0][1], t.normals[0][2], bits(ti)));
                cold.push_back(f4(t.normals[1][0], t.normals[1][1], t.normals[1][2], bits(t.material)));
                cold.push_back(f4(t.normals[2][0], t.normals[2][1], t.normals[2][2], 0.0f));
                double l1 = std::sqrt((double)e1[0] * e1[0] + (double)e1[1] * e1[1] + (double)e1[2] * e1[2]);
                double l2 = std::sqrt((double)e2[0] * e2[0] + (double)e2[1] * e2[1] + (double)e2[2] * e2[2]);
                float prod = (float)(l1 * l2), sum = (float)(l1 + l2);
                if (!(prod <= out->max_edge_product)) out->max_edge_product = prod;      // NaN sticks
                if (!(sum <= out->max_edge_sum)) out->max_edge_sum = sum;
            }
            ref = RT_REF_LEAF_BIT | (n.triangle_count << RT_REF_COUNT_SHIFT) | (uint32_t)first;
            if (n.triangle_count) {
                leaf_boxes.push_back(f4(n.aabb_min[0], n.aabb_min[1], n.aabb_min[2], n.aabb_max[0]));
                leaf_boxes.push_back(f4(n.aabb_max[1], n.aabb_max[2], bits(ref), 0.0f));
            }
        } else {
            if (!check_node(n.child_left) || !check_node(n.child_right))
                return fail(RT_ERR_OUT_OF_RANGE, "mesh %u: node %u has a child outside the %u nodes", mesh_index, it.src,
                            desc->n_mesh_bvh_nodes);
            ref = (uint32_t)(nodes.size() / 4);
            const RtBvhNode& l = desc->mesh_bvh_nodes[n.child_left];
            const RtBvhNode& r = desc->mesh_bvh_nodes[n.child_right];
            if (!inside(l.aabb_min, l.aabb_max, n.aabb_min, n.aabb_max) || !inside(r.aabb_min, r.aabb_max, n.aabb_min, n.aabb_max))
                out->nested = false;
            nodes.push_back(f4(l.aabb_min[0], l.aabb_min[1], l.aabb_min[2], l.aabb_max[0]));
            nodes.push_back(f4(l.aabb_max[1], l.aabb_max[2], r.aabb_min[0], r.aabb_min[1]));
            nodes.push_back(f4(r.aabb_min[2], r.aabb_max[0], r.aabb_max[1], r.aabb_max[2]));
            nodes.push_back(f4(bits(RT_REF_NONE), bits(RT_REF_NONE), 0.0f, 0.0f));
            stack.push_back({n.child_left, ref, false, it.depth + 1});
            stack.push_back({n.child_right, ref, true, it.depth + 1});
            // a depth-first walk in any child order holds at most one pending sibling per level
            if (it.depth + 2 > need) need = it.depth + 2;
        }
        if (it.parent == 0xFFFFFFFFu) {
            out->root_ref = ref;
            memcpy(out->root_lo, n.aabb_min, 12);
            memcpy(out->root_hi, n.aabb_max, 12);
        } else {
            float4& p = nodes[4 * (size_t)it.parent + 3];
            if (it.right) p.y = bits(ref); else p.x = bits(ref);
        }
    }
    out->stack_need = need;
    out->leaf_count = (uint32_t)(leaf_boxes.size() / 2) - out->leaf_begin;
    return RT_OK;
}

// Folds every other level of a mesh's binary tree away: the 4-wide node of binary inner node N holds, for each of
// N's two children, the child itself when it is a leaf and otherwise the child's own two children - two to four
// entries, each with its box, in the order the reference's walk would reach them (right before left at both levels,
// intersect.rs:253-256).  A ray passes the box of a folded-away child whenever it passes a box inside it (the slab test
// of intersect.rs:185-226 is monotone in the box), so a walk of the 4-wide tree tests exactly the leaves the
// reference's walk tests.  A child is only folded away when both of its children's boxes do lie inside its own - always,
// in a tree the reference's builder made; in a tree that breaks the rule such a child stays an entry with its own box
// (the walk then tests that box as the reference does) and a 4-wide node of its own.
// `nodes` is the binary layout of layout_mesh; leaves keep their slots.
static int collapse_mesh(const std::vector<float4>& nodes, uint32_t mesh_index, MeshLayout* ml, std::vector<float4>& nodes4) {
    ml->root_ref4 = ml->root_ref;
    ml->stack_need4 = 1;
    if (ml->root_ref == RT_REF_NONE || (ml->root_ref & RT_REF_LEAF_BIT)) return RT_OK;
    auto word = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    struct Entry { float lo[3], hi[3]; uint32_t ref; };
    // the two children of binary node b, right first
    auto children = [&](uint32_t b, Entry out[2]) {
        const float4 &a = nodes[4 * (size_t)b], &c = nodes[4 * (size_t)b + 1], &d = nodes[4 * (size_t)b + 2], &e = nodes[4 * (size_t)b + 3];
        Entry l = {{a.x, a.y, a.z}, {a.w, c.x, c.y}, word(e.x)}, r = {{c.z, c.w, d.x}, {d.y, d.z, d.w}, word(e.y)};
        out[0] = r; out[1] = l;
    };
    struct Job { uint32_t binary; uint32_t slot4; uint32_t pending; uint32_t depth; };
    std::vector<Job> jobs;
    auto new_node = [&]() { const uint32_t k = (uint32_t)(nodes4.size() / 8); nodes4.resize(nodes4.size() + 8, f4(0, 0, 0, 0)); return k; };
    ml->root_ref4 = new_node();
    jobs.push_back({ml->root_ref, ml->root_ref4, 0, 0});
    while (!jobs.empty()) {
        const Job job = jobs.back();
        jobs.pop_back();
        if (job.depth > 2 * RT_BVH_STACK_LEN) return fail(RT_ERR_INVALID_ARGUMENT, "mesh %u: BVH is not a tree", mesh_index);
        Entry two[2], entries[4];
        uint32_t n = 0;
        children(job.binary, two);
        for (int side = 0; side < 2; ++side) {
            if (two[side].ref == RT_REF_NONE) continue;
            if (two[side].ref & RT_REF_LEAF_BIT) { entries[n++] = two[side]; continue; }
            Entry grand[2];
            children(two[side].ref, grand);
            bool contained = true;
            for (int g = 0; g < 2; ++g)
                for (int k = 0; k < 3; ++k)
                    if (grand[g].ref != RT_REF_NONE && !(grand[g].lo[k] >= two[side].lo[k] && grand[g].hi[k] <= two[side].hi[k])) contained = false;
            if (!contained) { entries[n++] = two[side]; continue; }             // not folded: the entry leads to the child's own 4-wide node
            for (int g = 0; g < 2; ++g)
                if (grand[g].ref != RT_REF_NONE) entries[n++] = grand[g];
        }
        // a walk that goes on with one entry keeps the others on its stack
        const uint32_t pending = job.pending + (n ? n - 1 : 0);
        if (pending + 1 > ml->stack_need4) ml->stack_need4 = pending + 1;
        uint32_t refs[4] = {RT_REF_NONE, RT_REF_NONE, RT_REF_NONE, RT_REF_NONE};
        float box[4][6];
        for (uint32_t c = 0; c < 4; ++c)
            for (int k = 0; k < 6; ++k) box[c][k] = 0.0f;
        for (uint32_t c = 0; c < n; ++c) {
            for (int k = 0; k < 3; ++k) { box[c][k] = entries[c].lo[k]; box[c][3 + k] = entries[c].hi[k]; }
            if (entries[c].ref & RT_REF_LEAF_BIT) refs[c] = entries[c].ref;
            else {
                refs[c] = new_node();
                jobs.push_back({entries[c].ref, refs[c], pending, job.depth + 1});
            }
        }
        float4* q = &nodes4[8 * (size_t)job.slot4];
        // two boxes per three float4, as in the binary layout
        q[0] = f4(box[0][0], box[0][1], box[0][2], box[0][3]); q[1] = f4(box[0][4], box[0][5], box[1][0], box[1][1]);
        q[2] = f4(box[1][2], box[1][3], box[1][4], box[1][5]);
        q[3] = f4(box[2][0], box[2][1], box[2][2], box[2][3]); q[4] = f4(box[2][4], box[2][5], box[3][0], box[3][1]);
        q[5] = f4(box[3][2], box[3][3], box[3][4], box[3][5]);
        q[6] = f4(bits(refs[0]), bits(refs[1]), bits(refs[2]), bits(refs[3]));
    }
    return RT_OK;
}

// World-space culling box of an instance, see DInstance.  Every hit the reference can report on the instance
// lies, in mesh-local space, inside the root box (up to float rounding), and its world position is
// model * (that point +- 1e-4 * unit normal) (intersect.rs:312-316, 379-385).  The box is the model transform
// of the root box's corners, padded by far more than those effects: 1/128 of its extent, 1/4096 of its
// largest coordinate and twice the normal offset.  It is only trusted when the four matrices are finite,
// moderate in size and consistent with each other (model_inverse * model = I to 1e-5), and only for rays
// whose origin is within 4096 extents, so that the rounding of the reference's own local ray
// (relative 2^-23 of the origin's size) stays below the padding as well.
static void instance_world_box(const RtMeshInstance& in, const MeshLayout& ml, double fro_rs, DInstance* o) {
    o->cullable = 0;
    o->cull_origin_limit = 0.0f;
    for (int c = 0; c < 3; ++c) { o->world_lo[c] = 0.0f; o->world_hi[c] = 0.0f; }
    const float* mats[4] = {in.model, in.model_inverse, in.rotation_scale, in.rotation_scale_inverse};
    for (const float* m : mats)
        for (int k = 0; k < 16; ++k)
            if (!(std::fabs((double)m[k]) <= 1e15)) return;                       // also catches NaN
    auto consistent = [](const float* a, const float* b) {
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                double v = 0.0;
                for (int k = 0; k < 4; ++k) v += (double)a[4 * k + r] * (double)b[4 * c + k];
                if (!(std::fabs(v - (r == c ? 1.0 : 0.0)) <= 1e-5)) return false;
            }
        return true;
    };
    if (!consistent(in.model_inverse, in.model) || !consistent(in.rotation_scale_inverse, in.rotation_scale)) return;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int corner = 0; corner < 8; ++corner) {
        double p[3];
        for (int c = 0; c < 3; ++c) p[c] = (corner >> c & 1) ? (double)ml.root_hi[c] : (double)ml.root_lo[c];
        if (!(std::fabs(p[0]) <= 1e15 && std::fabs(p[1]) <= 1e15 && std::fabs(p[2]) <= 1e15)) return;
        for (int r = 0; r < 3; ++r) {
            double w = (double)in.model[12 + r];
            for (int c = 0; c < 3; ++c) w += (double)in.model[4 * c + r] * p[c];
            if (w < lo[r]) lo[r] = w;
            if (w > hi[r]) hi[r] = w;
        }
    }
    double ext = 0.0, mag = 0.0;
    for (int c = 0; c < 3; ++c) {
        if (!(lo[c] <= hi[c])) return;
        ext = std::max(ext, hi[c] - lo[c]);
        mag = std::max(mag, std::max(std::fabs(lo[c]), std::fabs(hi[c])));
    }
    if (!(ext > 0.0) || !(mag <= 1e15)) return;
    const double pad = ext / 128.0 + mag / 4096.0 + 2e-4 * fro_rs;
    for (int c = 0; c < 3; ++c) {
        o->world_lo[c] = std::nextafter((float)(lo[c] - pad), -INFINITY);
        o->world_hi[c] = std::nextafter((float)(hi[c] + pad), INFINITY);
    }
    o->cull_origin_limit = (float)(4096.0 * ext);
    o->cullable = 1;
}

extern "C" int rt_scene_create(const RtSceneDesc* desc, int device_id, RtScene** out_scene) {
    if (!desc || !out_scene) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: null argument");
    *out_scene = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) return fail(RT_ERR_NO_DEVICE, "no HIP device is visible");
    if (device_id < 0 || device_id >= n_dev) return fail(RT_ERR_NO_DEVICE, "device %d of %d does not exist", device_id, n_dev);
    struct { const void* p; uint32_t n; const char* name; } arrays[] = {
        {desc->lights, desc->n_lights, "lights"}, {desc->materials, desc->n_materials, "materials"},
        {desc->planes, desc->n_planes, "planes"}, {desc->spheres, desc->n_spheres, "spheres"},
        {desc->triangles, desc->n_triangles, "triangles"}, {desc->mesh_triangles, desc->n_mesh_triangles, "mesh_triangles"},
        {desc->mesh_bvh_nodes, desc->n_mesh_bvh_nodes, "mesh_bvh_nodes"}, {desc->meshes, desc->n_meshes, "meshes"},
        {desc->mesh_instances, desc->n_mesh_instances, "mesh_instances"}};
    for (auto& a : arrays)
        if (a.n && !a.p) return fail(RT_ERR_INVALID_ARGUMENT, "%s: %u elements but a null pointer", a.name, a.n);
    if (desc->n_mesh_instances > RT_MAX_INSTANCES)
        return fail(RT_ERR_OUT_OF_RANGE, "%u mesh instances exceed the limit of %u", desc->n_mesh_instances, RT_MAX_INSTANCES);
    if (desc->background.type > RT_BACKGROUND_COLORED_DIRECTION) return fail(RT_ERR_INVALID_ARGUMENT, "unknown background type %u", desc->background.type);

    // ---- flatten on the host --------------------------------------------------------------------
    std::vector<DLight> lights(desc->n_lights);
    for (uint32_t i = 0; i < desc->n_lights; ++i) {
        const RtLight& l = desc->lights[i];
        DLight& o = lights[i];
        memcpy(o.pos, l.position, 16);
        for (int c = 0; c < 3; ++c) o.world[c] = l.position[c] / l.position[3];   // raytracer.rs:162
        memcpy(o.ambient, l.ambient, 12); memcpy(o.diffuse, l.diffuse, 12); memcpy(o.specular, l.specular, 12);
    }
    std::vector<RtMaterial> materials(desc->materials, desc->materials + desc->n_materials);
    for (uint32_t i = 0; i < desc->n_materials; ++i)
        if (materials[i].material_type > RT_MATERIAL_REFLECT_AND_REFRACT)
            return fail(RT_ERR_INVALID_ARGUMENT, "material %u: unknown material_type %u", i, materials[i].material_type);
    // what a shadow verdict can change in radiance() per (material, light), see shadow_is_moot (RT_TRACE_MOOT_SHADOW_RAYS=1:
    // every shadow ray is traced, as a cross-check)
    std::vector<uint32_t> moot_flags((size_t)desc->n_materials * desc->n_lights, 0u);
    const char* trace_moot = getenv("RT_TRACE_MOOT_SHADOW_RAYS");
    if (!(trace_moot && atoi(trace_moot)))
        for (uint32_t m = 0; m < desc->n_materials; ++m)
            for (uint32_t l = 0; l < desc->n_lights; ++l) {
                const RtMaterial& mat = materials[m];
                bool d_zero = true, d_finite = true, s_zero = true, s_finite = true;
                for (int c = 0; c < 3; ++c) {
                    const float d = lights[l].diffuse[c] * mat.diffuse[c], sp = lights[l].specular[c] * mat.specular[c];   // as cmul in phong_pass
                    d_zero = d_zero && d == 0.0f;  d_finite = d_finite && std::isfinite(d);
                    s_zero = s_zero && sp == 0.0f; s_finite = s_finite && std::isfinite(sp);
                }
                moot_flags[(size_t)m * desc->n_lights + l] = (d_zero ? RT_MOOT_DIFFUSE_ZERO : 0u) | (d_finite ? RT_MOOT_DIFFUSE_FINITE : 0u) |
                                                             (s_zero && mat.shininess >= 0.0f ? RT_MOOT_SPECULAR_ZERO : 0u) |
                                                             (s_finite && mat.shininess > 0.0f ? RT_MOOT_SPECULAR_FINITE : 0u);
            }
    std::vector<DPlane> planes(desc->n_planes);
    for (uint32_t i = 0; i < desc->n_planes; ++i) {
        const RtPlane& p = desc->planes[i];
        if (p.material >= desc->n_materials) return fail(RT_ERR_OUT_OF_RANGE, "plane %u: material %u outside the %u materials", i, p.material, desc->n_materials);
        memcpy(planes[i].n, p.normal, 12); planes[i].d = p.distance; planes[i].material = p.material;
    }
    std::vector<DSphere> spheres(desc->n_spheres);
    for (uint32_t i = 0; i < desc->n_spheres; ++i) {
        const RtSphere& p = desc->spheres[i];
        if (p.material >= desc->n_materials) return fail(RT_ERR_OUT_OF_RANGE, "sphere %u: material %u outside the %u materials", i, p.material, desc->n_materials);
        memcpy(spheres[i].c, p.center, 12); spheres[i].r = p.radius; spheres[i].material = p.material;
    }
    std::vector<DTriangle> triangles(desc->n_triangles);
    for (uint32_t i = 0; i < desc->n_triangles; ++i) {
        const RtTriangle& t = desc->triangles[i];
        if (t.material >= desc->n_materials) return fail(RT_ERR_OUT_OF_RANGE, "triangle %u: material %u outside the %u materials", i, t.material, desc->n_materials);
        DTriangle& o = triangles[i];
        for (int c = 0; c < 3; ++c) {
            o.p0[c] = t.vertices[0][c];
            o.e1[c] = t.vertices[1][c] - t.vertices[0][c];
            o.e2[c] = t.vertices[2][c] - t.vertices[0][c];
            o.n0[c] = t.normals[0][c]; o.n1[c] = t.normals[1][c]; o.n2[c] = t.normals[2][c];
            o.fn[c] = t.face_normal[c];
        }
        o.material = t.material;
    }
    std::vector<float4> nodes, nodes4, hot, cold, leaf_boxes;
    std::vector<MeshLayout> layouts(desc->n_meshes);
    uint32_t stack_entries = 1, stack_entries4 = 1;
    for (uint32_t mi = 0; mi < desc->n_meshes; ++mi) {
        int rc = layout_mesh(desc, mi, nodes, hot, cold, leaf_boxes, &layouts[mi]);
        if (rc == RT_OK) rc = collapse_mesh(nodes, mi, &layouts[mi], nodes4);
        if (rc != RT_OK) return rc;
    }
    std::vector<DInstance> instances(desc->n_mesh_instances);
    for (uint32_t i = 0; i < desc->n_mesh_instances; ++i) {
        const RtMeshInstance& in = desc->mesh_instances[i];
        if (in.mesh >= desc->n_meshes) return fail(RT_ERR_OUT_OF_RANGE, "mesh instance %u: mesh %u outside the %u meshes", i, in.mesh, desc->n_meshes);
        if (in.material_override >= (int32_t)desc->n_materials || in.material_override < -1)
            return fail(RT_ERR_OUT_OF_RANGE, "mesh instance %u: material_override %d outside the %u materials", i, in.material_override, desc->n_materials);
        DInstance& o = instances[i];
        mat34(in.model_inverse, &o.model_inverse);
        mat34(in.rotation_scale_inverse, &o.rotation_scale_inverse);
        mat34(in.model, &o.model);
        mat34(in.rotation_scale, &o.rotation_scale);
        const MeshLayout& ml = layouts[in.mesh];
        memcpy(o.root_lo, ml.root_lo, 12);
        memcpy(o.root_hi, ml.root_hi, 12);
        o.root_ref = ml.root_ref;
        o.root_ref4 = ml.root_ref4;
        o.material_override = in.material_override;
        // Moeller-Trumbore's t carries a rounding error of at most about 10 eps K (t + D) / |a| with
        // K = |e1||e2|, D = |e1|+|e2| and |a| >= 1e-5 (intersect.rs:152-153): below 2^-7 (t + D) for
        // K <= 1/8, which the pruning slack of the kernel (2^-6 relative, 2^-5 D absolute) dominates.
        o.prunable = (ml.max_edge_product <= 0.125f) ? 1u : 0u;
        o.prune_abs = 0.03125f * ml.max_edge_sum;
        // A hit's world position is model * (local point +- 1e-4 * unit normal) (intersect.rs:312-316,
        // 379-385), so its world distance differs from (local t) / |rotation_scale_inverse * d| by at
        // most 1e-4 * ||rotation_scale|| plus rounding; the Frobenius norm bounds the spectral norm.
        double fro = 0.0;
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r) fro += (double)in.rotation_scale[4 * c + r] * in.rotation_scale[4 * c + r];
        o.world_guard = (float)(1.5e-4 * std::sqrt(fro));
        if (!(o.world_guard == o.world_guard) || o.world_guard > 1e30f) o.prunable = 0;
        instance_world_box(in, ml, std::sqrt(fro), &o);
        if (!ml.nested) { o.prunable = 0; o.cullable = 0; }        // see MeshLayout::nested: every ray takes the exact set-up, walks are not pruned
        // Far-away rays (DInstance::heavy_origin): the slab test's products carry an error of about 2^-22 of the origin's
        // coordinate, so from 2^19 root-box extents out a ray looks an eighth of the mesh thick on that axis; only worth a
        // block of its own for meshes with many leaves, and only exact for trees that nest
        o.leaf_begin = ml.leaf_begin;
        o.leaf_count = ml.leaf_count;
        o.heavy_origin = INFINITY;
        if (ml.nested && o.prunable && ml.leaf_count >= 1024) {
            float ext = 0.0f;
            for (int c = 0; c < 3; ++c) ext = std::max(ext, ml.root_hi[c] - ml.root_lo[c]);
            if (ext > 0.0f && ext < 1e15f) o.heavy_origin = ext * 524288.0f;
        }
        if (ml.stack_need > stack_entries) stack_entries = ml.stack_need;
        if (ml.stack_need4 > stack_entries4) stack_entries4 = ml.stack_need4;
    }
    if (stack_entries > RT_BVH_STACK_LEN)
        return fail(RT_ERR_BVH_TOO_DEEP, "a mesh BVH can need %u traversal stack entries; the reference's ArrayVec holds %d (intersect.rs:240)",
                    stack_entries, RT_BVH_STACK_LEN);

    // ---- upload ---------------------------------------------------------------------------------
    HIP_TRY(hipSetDevice(device_id));
    RtScene* s = new RtScene();
    s->device = device_id;
    int rc = RT_OK;
    auto bail = [&](int code) { rt_scene_destroy(s); return code; };
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return bail(fail(RT_ERR_DEVICE, "hipGetDeviceProperties failed"));
    s->n_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&s->ev_start) != hipSuccess || hipEventCreate(&s->ev_stop) != hipSuccess ||
        hipEventCreate(&s->ev_trace_start) != hipSuccess || hipEventCreate(&s->ev_trace_stop) != hipSuccess)
        return bail(fail(RT_ERR_DEVICE, "stream / event creation failed"));
    s->d.background = desc->background;
    s->d.n_lights = desc->n_lights; s->d.n_planes = desc->n_planes; s->d.n_spheres = desc->n_spheres;
    s->d.n_triangles = desc->n_triangles; s->d.n_instances = desc->n_mesh_instances;
    s->d.stack_entries = stack_entries4;
    s->d.stack_lds = std::min<uint32_t>(stack_entries4, RT_STACK_LDS_MAX);
    // range tree over the instances' culling boxes (see DScene), for scenes where a linear scan per ray hurts
    std::vector<float4> range_nodes;
    std::vector<float> range_limits;
    uint32_t range_depth = 0;
    {
        size_t range_tree_min = RT_RANGE_TREE_MIN;
        if (const char* env = getenv("RT_RANGE_TREE_MIN")) { const int v = atoi(env); if (v >= 0) range_tree_min = (size_t)v; }
        bool usable = instances.size() > range_tree_min;
        for (const DInstance& di : instances) usable = usable && (di.cullable || di.root_ref == RT_REF_NONE);
        if (usable) {
            struct Build {
                const std::vector<DInstance>& inst; std::vector<float4>& nodes; std::vector<float>& limits; uint32_t depth = 0;
                // returns the node of the range [a, b)
                uint32_t run(uint32_t a, uint32_t b, uint32_t level, float lo[3], float hi[3], float* limit) {
                    if (level > depth) depth = level;
                    const uint32_t k = (uint32_t)limits.size();
                    limits.push_back(0.0f);
                    nodes.push_back(f4(0, 0, 0, 0)); nodes.push_back(f4(0, 0, 0, 0));
                    if (b - a == 1) {
                        const DInstance& di = inst[a];
                        const bool empty = di.root_ref == RT_REF_NONE;           // an empty mesh: never a pair
                        for (int c = 0; c < 3; ++c) { lo[c] = empty ? INFINITY : di.world_lo[c]; hi[c] = empty ? -INFINITY : di.world_hi[c]; }
                        *limit = empty ? INFINITY : di.cull_origin_limit;
                        nodes[2 * k] = f4(lo[0], lo[1], lo[2], bits(empty ? 0xFFFFFFFFu : (0x80000000u | a)));
                        nodes[2 * k + 1] = f4(hi[0], hi[1], hi[2], bits(k + 1));
                        limits[k] = *limit;
                        return k;
                    }
                    const uint32_t mid = a + (b - a) / 2;
                    float llo[3], lhi[3], rlo[3], rhi[3], ll, rl;
                    const uint32_t left = run(a, mid, level + 1, llo, lhi, &ll), right = run(mid, b, level + 1, rlo, rhi, &rl);
                    for (int c = 0; c < 3; ++c) { lo[c] = std::min(llo[c], rlo[c]); hi[c] = std::max(lhi[c], rhi[c]); }
                    *limit = std::min(ll, rl);
                    (void)left; (void)right;                                     // pre-order: the left child is node k + 1, the right one follows its subtree
                    nodes[2 * k] = f4(lo[0], lo[1], lo[2], bits(0u));
                    nodes[2 * k + 1] = f4(hi[0], hi[1], hi[2], bits((uint32_t)limits.size()));
                    limits[k] = *limit;
                    return k;
                }
            } build{instances, range_nodes, range_limits};
            float lo[3], hi[3], limit;
            build.run(0, (uint32_t)instances.size(), 1, lo, hi, &limit);
            range_depth = 0;                                          // (the walk keeps no stack, see rt_setup)
        }
    }
    s->d.n_range_nodes = (uint32_t)range_limits.size();
    s->d.n_tri_slots = (uint32_t)(hot.size() / 3);
    s->d.n_leaf_boxes = (uint32_t)(leaf_boxes.size() / 2);
    if (const char* env = getenv("RT_HEAVY_PAIRS")) if (!atoi(env)) for (DInstance& di : instances) di.heavy_origin = INFINITY;
    s->d.range_depth = range_depth;
    s->d.all_cullable = 1;
    for (const DInstance& di : instances) {
        if (di.root_ref == RT_REF_NONE) continue;                  // an empty mesh is never tested
        if (!di.cullable) s->d.all_cullable = 0;
    }
    if ((rc = upload(s, lights, &s->d.lights)) || (rc = upload(s, materials, &s->d.materials)) ||
        (rc = upload(s, planes, &s->d.planes)) || (rc = upload(s, spheres, &s->d.spheres)) ||
        (rc = upload(s, triangles, &s->d.triangles)) || (rc = upload(s, instances, &s->d.instances)) ||
        (rc = upload(s, range_nodes, &s->d.range_nodes)) || (rc = upload(s, range_limits, &s->d.range_limits)) ||
        (rc = upload(s, nodes, &s->d.nodes)) || (rc = upload(s, nodes4, &s->d.nodes4)) || (rc = upload(s, hot, &s->d.tri_hot)) || (rc = upload(s, cold, &s->d.tri_cold)) ||
        (rc = upload(s, leaf_boxes, &s->d.leaf_boxes)) || (rc = upload(s, moot_flags, &s->d.moot_flags)))
        return bail(rc);
    // six blocks per compute unit for the list kernels, which is what rt_shade - held to 80 registers - keeps resident
    // (per queued 1080p bunnies frame: rt_shade 0.796 ms with five blocks and five waves per SIMD, 0.757 with six and six)
    s->list_blocks = (uint32_t)s->n_cus * 6;
    // rt_shade sets up the rays it makes, block by block - unless the scene has so many instances that rays walk the range
    // tree: what a ray costs then varies too much for a block to be left alone with its own rays (measured: the 2.6 M- and
    // 10.5 M-triangle grids lose 2-3 %, every other workload gains 2-9 % per frame)
    s->fuse_setup = s->d.n_range_nodes == 0 ? 1u : 0u;
    if (const char* env = getenv("RT_FUSE_SETUP")) s->fuse_setup = atoi(env) ? 1u : 0u;
    if (const char* env = getenv("RT_LIST_BLOCKS_PER_CU")) { int v = atoi(env); if (v >= 1 && v <= 8) s->list_blocks = (uint32_t)(s->n_cus * v); }
    {
        // rt_setup's prefix tables and pair rings take more LDS per block than six blocks leave: its grid is what fits at once
        const int setup_bpc = rt_query_setup_blocks_per_cu(&s->d, s->list_blocks + 1);
        if (setup_bpc < 1) return bail(fail(RT_ERR_DEVICE, "the set-up kernel does not fit a compute unit"));
        s->setup_blocks = std::min<uint32_t>(s->list_blocks, (uint32_t)(s->n_cus * setup_bpc));
    }
    int bpc = rt_query_traverse_blocks_per_cu(s->list_blocks + 1, s->d.stack_lds);
    if (bpc < 1) return bail(fail(RT_ERR_DEVICE, "the traversal kernel does not fit a compute unit"));
    if (const char* env = getenv("RT_BLOCKS_PER_CU")) { int v = atoi(env); if (v >= 1 && v <= bpc) bpc = v; }
    s->traverse_blocks = (uint32_t)(s->n_cus * bpc);
    if (hipHostMalloc((void**)&s->h_overflow, 2 * sizeof(unsigned int), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&s->d_overflow, const_cast<unsigned int*>(s->h_overflow), 0) != hipSuccess)
        return bail(fail(RT_ERR_DEVICE, "cannot allocate the host-visible overflow flag"));
    // Scenes whose walk data (4-wide nodes + triangles) is far beyond the eight 4 MB L2s: neighbouring pairs walk the same
    // meshes, so the BVH walk hands them out in runs (2.6 M-triangle grid: 7.5 -> 6.2 ms per frame); cache-resident scenes
    // keep the interleaving that balances the waves (contiguous runs there: 1.24 -> 1.40 ms)
    // (0: never, 1: always, 2: launch by launch, where a wave gets several chunks - see rt_traverse)
    s->walk_contiguous = (nodes4.size() + hot.size()) * sizeof(float4) > (64u << 20) ? 1u : 2u;
    if (const char* env = getenv("RT_WALK_CONTIGUOUS")) { const int v = atoi(env); if (v >= 0 && v <= 2) s->walk_contiguous = (uint32_t)v; }
    // The packet walk (rt_traverse_packet) is measured slower than the lane-by-lane walk at every depth of every workload
    // tried, the coherent first two included, so no launch uses it unless asked to.
    if (const char* env = getenv("RT_WALK_PACKET_LEVELS")) { const int v = atoi(env); if (v >= 0 && v <= RT_MAX_DEPTH_LIMIT + 1) s->walk_packet_levels = (uint32_t)v; }
    s->h_overflow[0] = 0;
    s->h_overflow[1] = 0;
    // In a scene whose instances all have culling boxes only stray rays (non-finite or huge components) need the exact
    // path - none at all in ordinary scenes: the kernel is left out until a frame turns out to need it.
    s->slow_enabled = !s->d.all_cullable;
    if (const char* env = getenv("RT_ALWAYS_LAUNCH_SLOW")) s->slow_enabled = s->slow_enabled || atoi(env) != 0;
    if ((rc = scratch_for_device(device_id, &s->scratch))) return bail(rc);
    *out_scene = s;
    return RT_OK;
}

extern "C" void rt_scene_destroy(RtScene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->own_stream) (void)hipStreamSynchronize(s->own_stream);
    if (s->last_stream) (void)hipStreamSynchronize(s->last_stream);
    for (void* p : s->scene_allocs) (void)hipFree(p);
    for (void* p : s->frame_allocs) (void)hipFree(p);
    if (s->d_pixels) (void)hipFree(s->d_pixels);
    if (s->d_aux_prim_out) (void)hipFree(s->d_aux_prim_out);
    if (s->d_aux_inst_out) (void)hipFree(s->d_aux_inst_out);
    if (s->d_aux_flags_out) (void)hipFree(s->d_aux_flags_out);
    if (s->d_aux_t_out) (void)hipFree(s->d_aux_t_out);
    if (s->h_overflow) (void)hipHostFree(const_cast<unsigned int*>(s->h_overflow));
    for (hipEvent_t e : s->marks) (void)hipEventDestroy(e);
    if (s->ev_start) (void)hipEventDestroy(s->ev_start);
    if (s->ev_stop) (void)hipEventDestroy(s->ev_stop);
    if (s->ev_trace_start) (void)hipEventDestroy(s->ev_trace_start);
    if (s->ev_trace_stop) (void)hipEventDestroy(s->ev_trace_stop);
    if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
    delete s;
}

extern "C" int rt_scene_set_views(RtScene* s, const float* screen_to_world, uint32_t n_views, uint32_t width, uint32_t height) {
    if (!s || !screen_to_world) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_set_views: null argument");
    if (width == 0 || height == 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_set_views: empty screen %ux%u", width, height);
    if (n_views == 0 || n_views > RT_MAX_VIEWS)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_set_views: %u views; between 1 and %d are supported", n_views, RT_MAX_VIEWS);
    s->views.assign(screen_to_world, screen_to_world + (size_t)n_views * 16);
    s->n_views = n_views;
    s->width = width;
    s->height = height;
    s->camera_set = true;
    return RT_OK;
}

extern "C" int rt_scene_set_camera(RtScene* s, const float screen_to_world[16], uint32_t width, uint32_t height) {
    if (!s || !screen_to_world) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_set_camera: null argument");
    if (width == 0 || height == 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_set_camera: empty screen %ux%u", width, height);
    return rt_scene_set_views(s, screen_to_world, 1, width, height);
}

static uint32_t bytes_per_pixel(uint32_t format) { return format == RT_FORMAT_RGBA8 ? 4u : 3u; }

struct Geometry { uint32_t tiles_x, tiles_y, shard_count, shard_index, tiles_per_shard, view_pixels, n_pixels; };

static int frame_geometry(const RtScene* s, const RtRenderParams* p, bool sharded, Geometry* g) {
    if (!s->camera_set) return fail(RT_ERR_INVALID_ARGUMENT, "the camera has not been set (rt_scene_set_camera)");
    if (p->output_format > RT_FORMAT_RGBA8) return fail(RT_ERR_INVALID_ARGUMENT, "unknown output_format %u", p->output_format);
    g->tiles_x = (s->width + RT_TILE_DIM - 1) / RT_TILE_DIM;
    g->tiles_y = (s->height + RT_TILE_DIM - 1) / RT_TILE_DIM;
    g->shard_count = (sharded && p->shard_count > 1) ? p->shard_count : 1;
    g->shard_index = (g->shard_count > 1) ? p->shard_index : 0;
    if (g->shard_index >= g->shard_count) return fail(RT_ERR_INVALID_ARGUMENT, "shard_index %u >= shard_count %u", g->shard_index, g->shard_count);
    uint64_t tiles = (uint64_t)g->tiles_x * g->tiles_y;
    g->tiles_per_shard = (uint32_t)((tiles + g->shard_count - 1) / g->shard_count);
    uint64_t view_pixels = (uint64_t)g->tiles_per_shard * RT_TILE_DIM * RT_TILE_DIM, n_pixels = view_pixels * s->n_views;
    if (n_pixels > RT_MAX_PIXELS_PER_SHARD)
        return fail(RT_ERR_INVALID_ARGUMENT, "%llu pixels per shard (%u views together) exceed the limit of %u",
                    (unsigned long long)n_pixels, s->n_views, RT_MAX_PIXELS_PER_SHARD);
    g->view_pixels = (uint32_t)view_pixels;
    g->n_pixels = (uint32_t)n_pixels;
    return RT_OK;
}

static uint32_t ceil_div(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// What a frame's buffers hold besides the pointers in DFrame.
struct FrameSizes { size_t pixel_clear_bytes = 0, counter_words = 0, stats_words = 0; };

// Entries per recursion depth of the three kinds of list: rays (the child rays made at the depth above), walk items ((ray,
// instance) pairs for the BVH walk) and hit records (whose shadow rays the next depth traces).
struct ListCaps { uint64_t rays = 0, fat = 0, hits = 0; };

// The first guess for a handle that has not measured anything: `rays` per depth, half as many walk items and hit records.
static ListCaps caps_from_rays(uint64_t rays) { ListCaps c; c.rays = rays; c.fat = rays / 2; c.hits = rays / 2; return c; }

// Lays out the per-frame buffers of DFrame for `n_pixels` local pixels, lists of `cap` entries per recursion depth cut
// into `blocks` producer segments (none shorter than seg_min) plus the shared overflow segment.  Memory comes from
// `take(bytes, &ptr)`, which returns RT_OK or an error code: hipMalloc for the handle's own buffers, a bump pointer over
// the static scratch array for rt_render_device_noalloc.
template <class Take>
static int carve_frame(uint32_t n_pixels, uint32_t n_lights, uint32_t blocks, const ListCaps& cap, uint32_t seg_min, bool aux,
                       uint32_t spill_entries, uint32_t spill_lanes, Take&& take, DFrame* out, FrameSizes* sz) {
    DFrame& f = *out;
    memset(&f, 0, sizeof(f));
    f.n_seg = blocks + 1;
    auto seg = [&](uint64_t total) { return std::max<uint32_t>(ceil_div(total, blocks), seg_min); };
    // (what a block emits varies with what its chunks show - most for the walk items: on the 1080p bunnies frame the fullest
    // segment holds 3.8 times the mean, 2.3 times for the child rays, 1.2 times for the hit records)
    f.seg_rays = seg(cap.rays);          f.ovf_rays = (uint32_t)(cap.rays / 8 + 1024);
    f.seg_fat = seg(cap.fat);            f.ovf_fat = (uint32_t)(cap.fat / 4 + 1024);
    f.seg_hits = seg(cap.hits);          f.ovf_hits = (uint32_t)(cap.hits / 8 + 1024);
    f.extra_rays_cap = std::max<uint32_t>(f.seg_rays, 512);
    f.extra_hits_cap = std::max<uint32_t>(f.seg_hits, 256);
    const size_t ray_slots = (size_t)blocks * f.seg_rays + f.ovf_rays;
    const size_t fat_slots = (size_t)blocks * f.seg_fat + f.ovf_fat;
    const size_t hit_slots = (size_t)blocks * f.seg_hits + f.ovf_hits;
    const size_t all_ray_slots = ray_slots + hit_slots * n_lights;      // + the shadow rays, see DFrame::rays
    if (all_ray_slots >= RT_META_RAY_LIMIT) return fail(RT_ERR_OUT_OF_RANGE, "a frame needs more than 2^31 ray slots per depth");
    sz->stats_words = (size_t)std::max<uint32_t>(blocks * 4, spill_lanes / 64) * 8;      // a row per wave of the largest grid
    // colour sums and poison words in one allocation, all counters in another: two clears per frame
    sz->pixel_clear_bytes = (size_t)3 * n_pixels * sizeof(long long) + (size_t)n_pixels * sizeof(unsigned int);
    sz->counter_words = 3 * (RT_MAX_DEPTH_LIMIT + 2) + sz->stats_words + (size_t)RT_COUNT_ROWS * (RT_MAX_DEPTH_LIMIT + 3) * f.n_seg;
    unsigned int* counters = nullptr;
    int rc;
    auto get = [&](size_t count, auto** p) {
        void* v = nullptr;
        int r = take(count * sizeof(**p), &v);
        *p = static_cast<std::remove_reference_t<decltype(*p)>>(v);
        return r;
    };
    if ((rc = get(sz->pixel_clear_bytes / sizeof(long long) + 1, &f.acc)) || (rc = get(sz->counter_words, &counters)) ||
        (rc = get((size_t)3 * n_pixels, &f.acc_big)) ||
        (rc = get(2 * ray_slots, &f.rays[0])) || (rc = get(2 * ray_slots, &f.rays[1])) ||      // shadow rays are not stored
        (rc = get(all_ray_slots, &f.best[0])) || (rc = get(all_ray_slots, &f.best[1])) ||
        (rc = get(fat_slots, &f.item_results[0])) || (rc = get(fat_slots, &f.item_results[1])) ||
        (rc = get(4 * fat_slots, &f.fat_items[0])) || (rc = get(4 * fat_slots, &f.fat_items[1])) || (rc = get(ray_slots, &f.slow_rays)) ||
        (rc = get(3 * hit_slots, &f.hits[0])) || (rc = get(3 * hit_slots, &f.hits[1])) ||
        (rc = get((size_t)(RT_MAX_DEPTH_LIMIT + 2) * RT_HEAVY_CAP, &f.heavy_items)) ||
        (rc = get((size_t)blocks * (f.extra_rays_cap + f.extra_hits_cap), &f.extras)))
        return rc;
    f.shadow_base = (uint32_t)ray_slots;
    f.n_fat_slots = (uint32_t)fat_slots;
    f.spill_lanes = spill_lanes;
    if (spill_entries && (rc = get((size_t)spill_entries * spill_lanes, &f.spill))) return rc;
    unsigned int* c = counters;
    // cleared per frame from the first word on: the per-depth counters, the statistics, then the count rows of the depths
    // the frame can reach (enqueue_frame clears no further)
    f.chunk_counter = c; c += RT_MAX_DEPTH_LIMIT + 2;
    f.slow_flags = c; c += RT_MAX_DEPTH_LIMIT + 2;
    f.heavy_count = c; c += RT_MAX_DEPTH_LIMIT + 2;
    f.wave_stats = c; c += sz->stats_words;
    f.counts = c;
    if (aux && ((rc = get((size_t)n_pixels, &f.aux_flags)) || (rc = get((size_t)n_pixels, &f.aux_prim)) ||
                (rc = get((size_t)n_pixels, &f.aux_inst)) || (rc = get((size_t)n_pixels, &f.aux_t))))
        return rc;
    return RT_OK;
}

// The fields of a frame that do not depend on where its buffers live.
static void fill_frame(const RtScene* s, const Geometry& g, uint32_t max_depth, DFrame* f) {
    // [3][n_pixels] colour sums, then [n_pixels] poison words, in the one allocation that is cleared per frame
    f->poison = reinterpret_cast<unsigned int*>(f->acc + (size_t)3 * g.n_pixels);
    f->width = s->width; f->height = s->height;
    f->tiles_x = g.tiles_x; f->tiles_y = g.tiles_y;
    f->shard_index = g.shard_index; f->shard_count = g.shard_count;
    f->n_local_tiles = g.tiles_per_shard; f->view_pixels = g.view_pixels; f->n_views = s->n_views; f->n_pixels = g.n_pixels;
    f->max_depth = max_depth;
    f->overflow = s->d_overflow;
    f->needs_slow = s->d_overflow + 1;
    f->slow_launched = s->slow_enabled ? 1u : 0u;
    f->walk_contiguous = s->walk_contiguous;
    f->walk_packet_levels = s->walk_packet_levels;
    f->fuse_setup = s->fuse_setup;
}

// Stack entries per walking lane kept in device memory: what a walk can need beyond the entries in LDS - and, where packet
// walks are in use, what a lane may put aside during a packet (RT_PACKET_ASIDE in rt_kernels.hip).
static uint32_t spill_entries(const RtScene* s) {
    return s->d.stack_entries - s->d.stack_lds + (s->walk_packet_levels ? 32u : 0u);
}

// Bytes the buffers of such a frame take (the layout run without memory behind it).
static size_t frame_bytes_for(const RtScene* s, uint32_t n_pixels, const ListCaps& cap, uint32_t seg_min, bool aux) {
    size_t used = 0;
    auto take = [&](size_t bytes, void** p) { *p = nullptr; used += (bytes + 255) & ~(size_t)255; return (int)RT_OK; };
    DFrame f;
    FrameSizes sz;
    if (carve_frame(n_pixels, s->d.n_lights, s->list_blocks, cap, seg_min, aux, spill_entries(s), s->traverse_blocks * 256u, take, &f, &sz))
        return ~(size_t)0;
    return used;
}

static int ensure_frame(RtScene* s, const Geometry& g, uint32_t max_depth, bool aux) {
    const uint32_t blocks = s->list_blocks;
    const bool forced = s->debug_queue_capacity != 0;
    const uint64_t limit = 1ull << 30;
    // A block's own segment is never smaller than 1024 entries: with small frames the work is spread unevenly
    // over the blocks, and whatever does not fit a segment has to go through the shared overflow segment.
    const uint32_t seg_min = forced ? 1u : 1024u;                            // (tests force overflows with tiny lists)
    // What the lists of one depth may hold.  A handle that has had its lists counted (measure_lists) gets half as much
    // again as its fullest depth held per pixel (the 1080p bunnies frame: 0.7 child rays, 0.5 walk items and 0.9 hit
    // records) - and room for the primary rays, one per pixel, which rt_generate deals out a chunk per wave at a time (a
    // block's share is up to an eighth above the mean at that size); before that, four rays per pixel - less where that
    // would not fit the device's free memory.  A frame that needs more has its lists doubled, see note_overflow.
    ListCaps want = caps_from_rays((uint64_t)4 * g.n_pixels);
    long env_capacity = 0;
    if (const char* env = getenv("RT_QUEUE_CAPACITY")) env_capacity = atol(env);
    if (forced) want = caps_from_rays(s->debug_queue_capacity);
    else if (env_capacity > 0) want = caps_from_rays((uint64_t)env_capacity);
    else if (s->seen_valid) {
        auto half_again = [&](double per_pixel) { return (uint64_t)(1.5 * per_pixel * (double)g.n_pixels) + 4096; };
        want.rays = std::max(half_again(s->seen_per_pixel[0]), (uint64_t)g.n_pixels + g.n_pixels / 8 + 4096);
        want.fat = half_again(s->seen_per_pixel[1]);
        want.hits = half_again(s->seen_per_pixel[2]);
    } else {
        while (want.rays > limit && want.rays > g.n_pixels) want = caps_from_rays(want.rays / 2);      // (passes of a quarter of 2^30 pixels and more)
        size_t free_bytes = 0, total_bytes = 0;
        if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess) {
            const size_t room = (free_bytes + s->frame_bytes) / 10 * 8;
            while (want.rays > g.n_pixels && frame_bytes_for(s, g.n_pixels, want, seg_min, aux) > room) want = caps_from_rays(want.rays / 2);
        }
    }
    auto largest = [](const ListCaps& c) { return std::max(c.rays, std::max(c.fat, c.hits)); };
    auto shifted = [](ListCaps c, uint32_t by) { c.rays <<= by; c.fat <<= by; c.hits <<= by; return c; };
    // doubled once per overflow seen, but never beyond 2^30 entries per depth: a frame that needs more keeps overflowing and
    // is reported as such, the handle stays usable for smaller frames
    while (s->capacity_shift > 0 && (s->capacity_shift >= 32 || (largest(want) << s->capacity_shift) > limit)) --s->capacity_shift;
    want = shifted(want, s->capacity_shift);
    if (s->must_grow) {
        // a frame overflowed the lists it had - which may have been longer than this size asks for (see `fits`); which of
        // its lists it was is not known, so every one gets longer than it was
        while (!forced && (want.rays <= s->key.rays || want.fat <= s->key.fat || want.hits <= s->key.hits) && largest(want) < limit) {
            want = shifted(want, 1);
            ++s->capacity_shift;
        }
        s->must_grow = false;
    }
    if (!forced) { want.rays = std::max<uint64_t>(want.rays, 4096); want.fat = std::max<uint64_t>(want.fat, 4096); want.hits = std::max<uint64_t>(want.hits, 4096); }
    if (largest(want) > limit) return fail(RT_ERR_OUT_OF_RANGE, "a frame needs more than 2^30 list entries per depth");
    RtScene::FrameKey k;
    k.w = g.n_pixels; k.h = s->d.n_lights; k.shard_count = g.shard_count; k.max_depth = max_depth; k.aux = aux; k.blocks = blocks;
    k.rays = want.rays; k.fat = want.fat; k.hits = want.hits;
    k.debug = forced;
    // Buffers made for a larger frame serve a smaller one (fewer views, say) as they are: the lists are only longer
    // than needed, and the per-pixel arrays are addressed through the current pixel count below.  With a forced list
    // capacity (tests) the match has to be exact, or a test's short lists would silently be the previous long ones.
    bool fits = forced || s->key.debug ? (k.w == s->key.w && k.rays == s->key.rays && k.fat == s->key.fat && k.hits == s->key.hits)
                                       : (k.w <= s->key.w && k.rays <= s->key.rays && k.fat <= s->key.fat && k.hits <= s->key.hits);
    // Buffers made on the first guess are given back once a frame of this very size has been counted and needs less
    // than 70 % of them (a handle that alternates between frame sizes keeps the larger buffers).
    if (s->trim_pending) {
        if (fits && !forced && !s->key.debug && k.w == s->key.w && s->seen_pixels == k.w && !s->frame_allocs.empty() &&
            frame_bytes_for(s, g.n_pixels, want, seg_min, s->key.aux) < s->frame_bytes / 10 * 7)
            fits = false;
        s->trim_pending = false;
    }
    bool same = fits && k.h == s->key.h && k.blocks == s->key.blocks && (s->key.aux || !aux) && !s->frame_allocs.empty();
    if (!same) {
        HIP_TRY(hipDeviceSynchronize());
        for (void* p : s->frame_allocs) (void)hipFree(p);
        s->frame_allocs.clear();
        s->frame_bytes = 0;
        s->key = RtScene::FrameKey();
        FrameSizes sz;
        auto take = [&](size_t bytes, void** p) {
            HIP_TRY(hipMalloc(p, bytes));
            s->frame_allocs.push_back(*p);
            s->frame_bytes += bytes;
            return (int)RT_OK;
        };
        int rc = carve_frame(g.n_pixels, s->d.n_lights, blocks, want, seg_min, aux, spill_entries(s),
                             s->traverse_blocks * 256u, take, &s->f, &sz);
        if (rc) return rc;
        s->stats_words = sz.stats_words;
        s->counter_words = sz.counter_words;
        HIP_TRY(hipMemset(s->f.acc_big, 0, (size_t)3 * g.n_pixels * sizeof(double)));     // rt_resolve keeps it zero afterwards
        s->key = k;
    }
    fill_frame(s, g, max_depth, &s->f);
    s->pixel_clear_bytes = (size_t)3 * g.n_pixels * sizeof(long long) + (size_t)g.n_pixels * sizeof(unsigned int);
    return RT_OK;
}

// ---- the static scratch of rt_render_device_noalloc -------------------------------------------------------------
// One array per GPU the library is loaded on, held until the process ends.  Host code never uses the array's name
// as an address: rt_scene_create asks the runtime for this device's copy once (scratch_for_device) and everything
// else goes through that pointer.
#define RT_NOALLOC_SCRATCH_BYTES (128ull << 20)
#define RT_NOALLOC_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

static int scratch_for_device(int device, void** out) {
    static std::mutex lock;
    static void* table[RT_NOALLOC_MAX_DEVICES] = {};
    *out = nullptr;
    if (device < 0 || device >= RT_NOALLOC_MAX_DEVICES) return RT_OK;      // such a handle only lacks the entry point below
    std::lock_guard<std::mutex> hold(lock);
    if (!table[device]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));              // the device of the handle is current here
        table[device] = p;
    }
    *out = table[device];
    return RT_OK;
}

// The host sees a frame's overflow through the word the kernels store sequence numbers to.
static void note_overflow(RtScene* s) {
    const uint32_t slow = s->h_overflow[1];
    if (slow != s->slow_seen) {
        s->slow_seen = slow;
        s->overflow_unreported = true;
        s->slow_enabled = true;            // from now on the exact-path kernel runs at every depth
    }
    const uint32_t o = s->h_overflow[0];
    if (o != s->overflow_seen) {
        s->overflow_seen = o;
        s->overflow_unreported = true;
        // The handle's own lists get longer at the next frame - unless the frame that overflowed was one of
        // rt_render_device_noalloc, whose lists live in the static scratch and cannot grow: that is reported, nothing more.
        // (The word names the latest frame that overflowed; a frame of the handle's own lists that overflowed before it in
        // the same interval makes the next such frame overflow again, and is caught then.)
        const bool noalloc = o >= s->noalloc_seq_lo && o <= s->noalloc_seq_hi;
        if (!noalloc) {
            if (s->capacity_shift < RT_MAX_CAPACITY_SHIFT) ++s->capacity_shift;
            s->must_grow = true;
        }
    }
}

// Enqueues one frame on `stream`: clear, primary rays, the launches of every recursion depth (one more depth than
// max_depth: the shadow rays of the deepest hits), resolve into d_out.  `noalloc`: the frame's buffers are carved
// out of the static scratch instead of the handle's own allocations, and nothing here allocates.
static int enqueue_frame(RtScene* s, const RtRenderParams* p, bool sharded, bool aux, uint8_t* d_out, bool noalloc = false) {
    HIP_TRY(hipSetDevice(s->device));
    Geometry g;
    int rc = frame_geometry(s, p, sharded, &g);
    if (rc) return rc;
    uint32_t max_depth = p->max_depth ? p->max_depth : RT_DEFAULT_MAX_DEPTH;
    if (max_depth > RT_MAX_DEPTH_LIMIT) return fail(RT_ERR_INVALID_ARGUMENT, "max_depth %u exceeds the limit of %d", max_depth, RT_MAX_DEPTH_LIMIT);
    if (s->d.n_lights > 255) return fail(RT_ERR_INVALID_ARGUMENT, "%u lights; at most 255 are supported", s->d.n_lights);
    if (p->mode > RT_MODE_DEPTH_MAP || p->reserved != 0) return fail(RT_ERR_INVALID_ARGUMENT, "unknown mode %u / reserved %u", p->mode, p->reserved);
    note_overflow(s);                      // a list of an earlier frame did not fit: that frame was invalid, this one gets longer lists
    DFrame f;
    FrameSizes sz;
    uint32_t blocks = s->list_blocks;
    if (noalloc) {
        if (!s->scratch) return fail(RT_ERR_DEVICE, "rt_render_device_noalloc: device %d has no scratch array", s->device);
        // one list segment per compute unit, eight rays per pixel and depth (these lists cannot grow)
        blocks = (uint32_t)s->n_cus;
        size_t used = 0;
        auto take = [&](size_t bytes, void** ptr) {
            const size_t at = (used + 255) & ~(size_t)255;
            *ptr = static_cast<unsigned char*>(s->scratch) + at;           // never dereferenced beyond the limit checked below
            used = at + bytes;
            return (int)RT_OK;
        };
        const ListCaps cap = caps_from_rays(std::max<uint64_t>((uint64_t)8 * g.n_pixels, 4096));
        if ((rc = carve_frame(g.n_pixels, s->d.n_lights, blocks, cap, 64u, false, spill_entries(s),
                              s->traverse_blocks * 256u, take, &f, &sz))) return rc;
        if (used > RT_NOALLOC_SCRATCH_BYTES)
            return fail(RT_ERR_OUT_OF_RANGE,
                        "rt_render_device_noalloc: %u pixels with %u lights need %llu bytes of scratch, the static array holds %llu; "
                        "use rt_render_device for frames of this size", g.n_pixels, s->d.n_lights, (unsigned long long)used,
                        (unsigned long long)RT_NOALLOC_SCRATCH_BYTES);
        fill_frame(s, g, max_depth, &f);
    } else {
        if ((rc = ensure_frame(s, g, max_depth, aux))) return rc;
        f = s->f;
        sz.pixel_clear_bytes = s->pixel_clear_bytes; sz.counter_words = s->counter_words; sz.stats_words = s->stats_words;
    }
    hipStream_t stream = p->stream ? (hipStream_t)p->stream : s->own_stream;
    s->last_stream = stream;
    f.mode = p->mode;
    if (++s->frame_seq == 0) s->frame_seq = 1;
    f.overflow_seq = s->frame_seq;
    if (noalloc) {
        if (s->noalloc_seq_hi + 1 != s->frame_seq || s->noalloc_seq_lo > s->noalloc_seq_hi) s->noalloc_seq_lo = s->frame_seq;
        s->noalloc_seq_hi = s->frame_seq;
    }
    if (!aux) { f.aux_flags = f.aux_prim = f.aux_inst = nullptr; f.aux_t = nullptr; }
    HIP_TRY(hipEventRecord(s->ev_start, stream));
    HIP_TRY(hipMemsetAsync(f.acc, 0, sz.pixel_clear_bytes, stream));                              // colour sums + poison words
    // a depth map only looks at the primary hits
    const uint32_t last = p->mode == RT_MODE_DEPTH_MAP ? 0 : (s->d.n_lights > 0 ? max_depth : max_depth - 1);
    // every counter, the statistics and the count rows of depths 0 .. last + 1
    const size_t clear_words = 3 * (RT_MAX_DEPTH_LIMIT + 2) + sz.stats_words + (size_t)RT_COUNT_ROWS * (last + 2) * f.n_seg;
    HIP_TRY(hipMemsetAsync(f.chunk_counter, 0, clear_words * sizeof(unsigned int), stream));
    // the scratch array keeps what the last call on this GPU left in it
    if (noalloc) HIP_TRY(hipMemsetAsync(f.acc_big, 0, (size_t)3 * g.n_pixels * sizeof(double), stream));
    if (aux) HIP_TRY(hipMemsetAsync(f.aux_flags, 0, (size_t)g.n_pixels * sizeof(unsigned int), stream));
    HIP_TRY(hipEventRecord(s->ev_trace_start, stream));
    HIP_TRY(rt_launch_generate(&f, s->views.data(), blocks, stream));
    s->marked_levels = 0;
    if (s->kernel_timing && !noalloc) {
        while (s->marks.size() < (RT_N_LEVEL_KERNELS + 1) * (size_t)(last + 1)) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreate(&e));
            s->marks.push_back(e);
        }
        s->marked_levels = last + 1;
    }
    for (uint32_t level = 0; level <= last; ++level)
        HIP_TRY(rt_launch_level(&s->d, &f, level, last, blocks, std::min(blocks, s->setup_blocks), s->traverse_blocks, stream,
                                s->marked_levels ? &s->marks[(RT_N_LEVEL_KERNELS + 1) * (size_t)level] : nullptr));
    HIP_TRY(hipEventRecord(s->ev_trace_stop, stream));
    HIP_TRY(rt_launch_resolve(&f, d_out, bytes_per_pixel(p->output_format), g.shard_count > 1 ? 1u : 0u, stream));
    HIP_TRY(hipEventRecord(s->ev_stop, stream));
    s->timed = true;
    s->last_f = f;
    s->last_fused = f.fuse_setup != 0;
    s->last_stats_words = sz.stats_words;
    s->last_rows = last + 2;
    s->last_noalloc = noalloc;
    return RT_OK;
}

extern "C" int rt_shard_bytes(const RtScene* s, const RtRenderParams* p, uint64_t* out_bytes) {
    if (!s || !p || !out_bytes) return fail(RT_ERR_INVALID_ARGUMENT, "rt_shard_bytes: null argument");
    Geometry g;
    int rc = frame_geometry(s, p, true, &g);
    if (rc) return rc;
    if (g.shard_count > 1) *out_bytes = (uint64_t)g.view_pixels * bytes_per_pixel(p->output_format);
    else *out_bytes = (uint64_t)s->width * s->height * bytes_per_pixel(p->output_format);
    return RT_OK;
}

extern "C" int rt_render_device(RtScene* s, const RtRenderParams* p, void* d_out) {
    if (!s || !p || !d_out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_device: null argument");
    return enqueue_frame(s, p, true, false, static_cast<uint8_t*>(d_out));
}

extern "C" int rt_render_device_noalloc(RtScene* s, const RtRenderParams* p, void* d_out) {
    if (!s || !p || !d_out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_device_noalloc: null argument");
    return enqueue_frame(s, p, true, false, static_cast<uint8_t*>(d_out), true);
}

extern "C" int rt_noalloc_scratch_bytes(uint64_t* out_bytes) {
    if (!out_bytes) return fail(RT_ERR_INVALID_ARGUMENT, "rt_noalloc_scratch_bytes: null argument");
    *out_bytes = RT_NOALLOC_SCRATCH_BYTES;
    return RT_OK;
}

static int measure_lists(RtScene* s);

extern "C" int rt_render(RtScene* s, const RtRenderParams* p, uint8_t* out_pixels, RtAux* aux) {
    if (!s || !p || !out_pixels) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: null argument");
    HIP_TRY(hipSetDevice(s->device));
    if (!s->camera_set) return fail(RT_ERR_INVALID_ARGUMENT, "the camera has not been set (rt_scene_set_camera)");
    size_t n = (size_t)s->width * s->height * s->n_views;
    size_t bytes = n * bytes_per_pixel(p->output_format);
    if (s->d_pixels_bytes < bytes) {
        if (s->d_pixels) (void)hipFree(s->d_pixels);
        s->d_pixels = nullptr; s->d_pixels_bytes = 0;
        HIP_TRY(hipMalloc((void**)&s->d_pixels, bytes));
        s->d_pixels_bytes = bytes;
    }
    bool want_aux = aux && (aux->prim_id || aux->instance_id || aux->flags || aux->hit_t);
    s->regrown = 0;
    note_overflow(s);
    const bool owed = s->overflow_unreported;      // an overflow of an earlier rt_render_device frame, still to be reported
    for (;;) {
        int rc = enqueue_frame(s, p, false, want_aux, s->d_pixels);
        if (rc) return rc;
        hipStream_t stream = s->last_stream;
        if (want_aux) {
            if (s->aux_out_pixels < n) {
                if (s->d_aux_prim_out) (void)hipFree(s->d_aux_prim_out);
                if (s->d_aux_inst_out) (void)hipFree(s->d_aux_inst_out);
                if (s->d_aux_flags_out) (void)hipFree(s->d_aux_flags_out);
                if (s->d_aux_t_out) (void)hipFree(s->d_aux_t_out);
                s->d_aux_prim_out = s->d_aux_inst_out = nullptr; s->d_aux_flags_out = nullptr; s->d_aux_t_out = nullptr; s->aux_out_pixels = 0;
                HIP_TRY(hipMalloc((void**)&s->d_aux_prim_out, n * 4));
                HIP_TRY(hipMalloc((void**)&s->d_aux_inst_out, n * 4));
                HIP_TRY(hipMalloc((void**)&s->d_aux_flags_out, n));
                HIP_TRY(hipMalloc((void**)&s->d_aux_t_out, n * 4));
                s->aux_out_pixels = n;
            }
            DFrame f = s->last_f;
            HIP_TRY(rt_launch_resolve_aux(&f, s->d_aux_prim_out, s->d_aux_inst_out, s->d_aux_flags_out, s->d_aux_t_out, stream));
            if (aux->prim_id) HIP_TRY(hipMemcpyAsync(aux->prim_id, s->d_aux_prim_out, n * 4, hipMemcpyDeviceToHost, stream));
            if (aux->instance_id) HIP_TRY(hipMemcpyAsync(aux->instance_id, s->d_aux_inst_out, n * 4, hipMemcpyDeviceToHost, stream));
            if (aux->flags) HIP_TRY(hipMemcpyAsync(aux->flags, s->d_aux_flags_out, n, hipMemcpyDeviceToHost, stream));
            if (aux->hit_t) HIP_TRY(hipMemcpyAsync(aux->hit_t, s->d_aux_t_out, n * 4, hipMemcpyDeviceToHost, stream));
        }
        HIP_TRY(hipMemcpyAsync(out_pixels, s->d_pixels, bytes, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        const uint32_t seen = s->overflow_seen, seen_slow = s->slow_seen;
        note_overflow(s);
        if (s->overflow_seen == seen && s->slow_seen == seen_slow) break;
        // a ray list did not fit: the lists get longer (note_overflow) and the frame is rendered again, which is all
        // the report this overflow needs
        s->overflow_unreported = owed;
        ++s->regrown;
        if (s->regrown > 16) return fail(RT_ERR_OUT_OF_RANGE, "the ray lists keep overflowing");
    }
    if (aux) {
        memset(aux->counters, 0, sizeof(aux->counters));
        return rt_last_ray_counts(s, aux->counters);
    }
    return measure_lists(s);
}

extern "C" int rt_assemble_shards(RtScene* s, const RtRenderParams* p, const void* d_gathered, void* d_frame) {
    if (!s || !p || !d_gathered || !d_frame) return fail(RT_ERR_INVALID_ARGUMENT, "rt_assemble_shards: null argument");
    HIP_TRY(hipSetDevice(s->device));
    Geometry g;
    int rc = frame_geometry(s, p, true, &g);
    if (rc) return rc;
    hipStream_t stream = p->stream ? (hipStream_t)p->stream : s->own_stream;
    uint32_t bpp = bytes_per_pixel(p->output_format);
    if (g.shard_count <= 1) {
        HIP_TRY(hipMemcpyAsync(d_frame, d_gathered, (size_t)s->width * s->height * bpp * s->n_views, hipMemcpyDeviceToDevice, stream));
        return RT_OK;
    }
    HIP_TRY(rt_launch_assemble(static_cast<const uint8_t*>(d_gathered), static_cast<uint8_t*>(d_frame), s->width, s->height,
                               g.tiles_x, g.shard_count, g.tiles_per_shard, bpp, s->n_views, stream));
    return RT_OK;
}

static int sum_wave_stats(RtScene* s, uint64_t sums[8]) {
    size_t n = s->last_stats_words;
    std::vector<unsigned int> host(n);
    HIP_TRY(hipMemcpy(host.data(), s->last_f.wave_stats, n * sizeof(unsigned int), hipMemcpyDeviceToHost));
    for (int k = 0; k < 8; ++k) sums[k] = 0;
    for (size_t i = 0; i < n; ++i) sums[i % 8] += host[i];
    return RT_OK;
}

// Counts what the lists of the frame enqueued last held at their fullest depth (the stream has been waited for) and, if the
// frame was complete, remembers it per pixel: the next buffers this handle makes are sized by it (ensure_frame).
static int measure_lists(RtScene* s) {
    if (s->last_noalloc || s->last_rows == 0) return RT_OK;               // lists in the static scratch are not the handle's to size
    const DFrame& f = s->last_f;
    const size_t row = f.n_seg, words = (size_t)RT_COUNT_ROWS * s->last_rows * row;
    std::vector<unsigned int> host(words);
    HIP_TRY(hipMemcpy(host.data(), f.counts, words * sizeof(unsigned int), hipMemcpyDeviceToHost));
    if (getenv("RT_DEBUG_LISTS")) {
        // per depth and list: entries in all, the fullest producer segment against its capacity, the shared overflow segment
        const char* names[RT_COUNT_ROWS] = {"rays", "hits", "walk", "slow"};
        const uint32_t seg_cap[RT_COUNT_ROWS] = {f.seg_rays, f.seg_hits, f.seg_fat, f.seg_rays};
        const uint32_t ovf_cap[RT_COUNT_ROWS] = {f.ovf_rays, f.ovf_hits, f.ovf_fat, f.ovf_rays};
        for (uint32_t d = 0; d < s->last_rows; ++d)
            for (int k = 0; k < RT_COUNT_ROWS; ++k) {
                const unsigned int* c = &host[((size_t)RT_COUNT_ROWS * d + k) * row];
                uint64_t sum = 0; unsigned int fullest = 0;
                for (size_t i = 0; i + 1 < row; ++i) { sum += c[i]; fullest = std::max(fullest, c[i]); }
                if (sum + c[row - 1]) fprintf(stderr, "[lists] frame %u depth %u %s: %llu entries, fullest segment %u of %u, overflow segment %u of %u\n",
                                              s->frame_seq, d, names[k], (unsigned long long)(sum + c[row - 1]), fullest, seg_cap[k], c[row - 1], ovf_cap[k]);
            }
    }
    const int rows_of[3] = {RT_ROW_RAYS, RT_ROW_FAT, RT_ROW_HITS};
    uint64_t child_rays = 0;                                               // the fullest ray list below the primary rays'
    for (int k = 0; k < 3; ++k) {
        uint64_t most = 0;
        for (uint32_t d = 0; d < s->last_rows; ++d) {
            uint64_t sum = 0;
            const unsigned int* c = &host[((size_t)RT_COUNT_ROWS * d + rows_of[k]) * row];
            for (size_t i = 0; i < row; ++i) sum += c[i];
            most = std::max(most, sum);
            if (k == 0 && d > 0) child_rays = std::max(child_rays, sum);
        }
        s->last_used[k] = most;
    }
    if (s->h_overflow[0] == s->frame_seq || s->debug_queue_capacity) return RT_OK;      // clipped counts say nothing
    const uint64_t sized_by[3] = {child_rays, s->last_used[1], s->last_used[2]};
    for (int k = 0; k < 3; ++k) s->seen_per_pixel[k] = std::max(s->seen_per_pixel[k], (double)sized_by[k] / (double)f.n_pixels);
    s->seen_valid = true;
    s->seen_pixels = f.n_pixels;
    s->trim_pending = true;
    if (!s->must_grow) s->capacity_shift = 0;                               // the measured need replaces the doublings
    return RT_OK;
}

extern "C" int rt_last_list_use(RtScene* s, uint64_t out[RT_N_LIST_USE]) {
    if (!s || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_last_list_use: null argument");
    if (!s->timed) return fail(RT_ERR_INVALID_ARGUMENT, "nothing has been rendered on this handle yet");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->last_stream));
    note_overflow(s);
    int rc = measure_lists(s);
    if (rc) return rc;
    out[RT_LIST_USE_BYTES] = s->frame_bytes;
    out[RT_LIST_USE_RAY_SLOTS] = s->key.rays;
    out[RT_LIST_USE_WALK_SLOTS] = s->key.fat;
    out[RT_LIST_USE_HIT_SLOTS] = s->key.hits;
    out[RT_LIST_USE_RAYS] = s->last_used[0];
    out[RT_LIST_USE_WALK_ITEMS] = s->last_used[1];
    out[RT_LIST_USE_HITS] = s->last_used[2];
    return RT_OK;
}

extern "C" int rt_last_ray_counts(RtScene* s, uint64_t counters[RT_N_COUNTERS]) {
    if (!s || !counters) return fail(RT_ERR_INVALID_ARGUMENT, "rt_last_ray_counts: null argument");
    if (!s->timed) return fail(RT_ERR_INVALID_ARGUMENT, "nothing has been rendered on this handle yet");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->last_stream));
    note_overflow(s);
    if (s->overflow_unreported) {
        s->overflow_unreported = false;
        return fail(RT_ERR_OUT_OF_RANGE, "a frame rendered since the last query is incomplete: a ray list overflowed (latest such frame: number %u of "
                    "this handle) or rays needed the exact path while it was switched off (latest: %u); %u frames have been enqueued, the next one "
                    "gets longer lists / the exact path", s->overflow_seen, s->slow_seen, s->frame_seq);
    }
    uint64_t sums[8];
    int rc = sum_wave_stats(s, sums);
    if (rc) return rc;
    if ((rc = measure_lists(s))) return rc;
    for (int k = 0; k < RT_N_COUNTERS; ++k) counters[k] = 0;
    // a wave's row: [0] shadow rays not traced, [1..4] primary, shadow, reflected and transmitted rays, [5..7] the BVH walk
    for (int k = 1; k < 5; ++k) { counters[k] = sums[k]; counters[RT_CTR_RAYS] += sums[k]; }
    counters[RT_CTR_SHADOW_NOT_TRACED] = sums[0];
    counters[RT_CTR_WALK_NODE_VISITS] = sums[5];
    counters[RT_CTR_WALK_TRIANGLE_TESTS] = sums[6];
    counters[RT_CTR_WALK_HEAVY_PAIRS] = sums[7];
    return RT_OK;
}

extern "C" int rt_last_render_ms(RtScene* s, float* out_ms) {
    if (!s || !out_ms) return fail(RT_ERR_INVALID_ARGUMENT, "rt_last_render_ms: null argument");
    if (!s->timed) return fail(RT_ERR_INVALID_ARGUMENT, "nothing has been rendered on this handle yet");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventSynchronize(s->ev_stop));
    HIP_TRY(hipEventElapsedTime(out_ms, s->ev_start, s->ev_stop));
    return RT_OK;
}

extern "C" int rt_last_trace_ms(RtScene* s, float* out_ms) {
    if (!s || !out_ms) return fail(RT_ERR_INVALID_ARGUMENT, "rt_last_trace_ms: null argument");
    if (!s->timed) return fail(RT_ERR_INVALID_ARGUMENT, "nothing has been rendered on this handle yet");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventSynchronize(s->ev_trace_stop));
    HIP_TRY(hipEventElapsedTime(out_ms, s->ev_trace_start, s->ev_trace_stop));
    return RT_OK;
}

extern "C" int rt_set_kernel_timing(RtScene* s, int enable) {
    if (!s) return fail(RT_ERR_INVALID_ARGUMENT, "rt_set_kernel_timing: null argument");
    s->kernel_timing = enable != 0;
    return RT_OK;
}

extern "C" int rt_last_kernel_ms(RtScene* s, float out_ms[RT_N_LEVEL_KERNELS], uint32_t out_launches[RT_N_LEVEL_KERNELS]) {
    if (!s || !out_ms || !out_launches) return fail(RT_ERR_INVALID_ARGUMENT, "rt_last_kernel_ms: null argument");
    if (!s->timed || s->marked_levels == 0)
        return fail(RT_ERR_INVALID_ARGUMENT, "the last frame was not rendered in kernel timing mode (rt_set_kernel_timing)");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventSynchronize(s->ev_stop));
    for (int k = 0; k < RT_N_LEVEL_KERNELS; ++k) { out_ms[k] = 0.0f; out_launches[k] = 0; }
    for (uint32_t level = 0; level < s->marked_levels; ++level) {
        const hipEvent_t* m = &s->marks[(RT_N_LEVEL_KERNELS + 1) * (size_t)level];
        for (int k = 0; k < RT_N_LEVEL_KERNELS; ++k) {
            const bool launched = (k == RT_KERNEL_SETUP && (level == 0 || !s->last_fused)) || k == RT_KERNEL_SHADE ||
                                  (k == RT_KERNEL_TRAVERSE && s->d.n_instances > 0);
            if (!launched) continue;
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, m[k], m[k + 1]));
            out_ms[k] += ms;
            out_launches[k] += 1;
        }
    }
    return RT_OK;
}

// Test aids: the first caps the ray lists of a recursion depth (0 restores the default of four rays per pixel), the
// second reports how often rt_render had to enlarge the lists and redo its last frame.
extern "C" int rt_debug_set_queue_capacity(RtScene* s, uint32_t capacity) {
    if (!s) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_set_queue_capacity: null argument");
    s->debug_queue_capacity = capacity;
    return RT_OK;
}

extern "C" int rt_debug_last_regrown(RtScene* s, uint32_t* out) {
    if (!s || !out) return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_last_regrown: null argument");
    if (!s->timed) return fail(RT_ERR_INVALID_ARGUMENT, "nothing has been rendered on this handle yet");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->last_stream));
    *out = s->regrown;
    return RT_OK;
}


extern "C" const char* rt_last_error(void) { return g_error; }

// shared with host/rt_host.cpp so that both halves of the library report through rt_last_error()
extern "C" char* rt_internal_error_buffer(void) { return g_error; }
