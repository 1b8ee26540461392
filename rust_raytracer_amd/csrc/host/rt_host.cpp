// Setup-time host code of the drop-in (include/rt_host.h): scene JSON -> arrays of RtSceneDesc.
//
// Mirrors, for callers without a Rust toolchain, what lib_raytracer does before the first frame:
// scene_file/parser.rs (JSON schema of scene_file/json_format.rs), object_file.rs (OBJ meshes via
// tobj's triangulate + single_index options), bvh/builder.rs (5-bin SAH, leaves of <= 5 triangles,
// right-first node order) and raytracing/transform.rs + instance.rs (camera and instance
// matrices with nalgebra's operation order).  The BVH this produces is the same tree, node for
// node, as the reference builder's; it is built with one pass per node instead of nine (bin hulls
// are merged instead of re-scanned - min/max are exact, so the merge order does not matter).
//
// Compiled with -ffp-contract=off: float arithmetic below is plain IEEE binary32.
#include <cctype>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <limits>
#include <map>
#include <string>
#include <vector>

#include "../../../include/rt_host.h"

extern "C" char* rt_internal_error_buffer(void);      // rt_api.hip: the calling thread's 512-byte error text

namespace {

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(rt_internal_error_buffer(), 512, fmt, ap);      // read back through rt_last_error()
    va_end(ap);
    return code;
}

struct Error {
    int code;
    std::string what;
};
[[noreturn]] void raise(int code, const std::string& what) { throw Error{code, what}; }

// ------------------------------------------------------------------------------------------------
// small linear algebra, column-major 4x4 (element (row, col) at [4*col + row])
// ------------------------------------------------------------------------------------------------
struct M4 {
    float a[16];
    float& at(int r, int c) { return a[4 * c + r]; }
    float at(int r, int c) const { return a[4 * c + r]; }
};

M4 identity() {
    M4 m{};
    for (int i = 0; i < 4; ++i) m.at(i, i) = 1.0f;
    return m;
}

// nalgebra multiplies statically sized matrices one result column at a time, each column as
// col0*x0, then += col1*x1, += col2*x2, += col3*x3
M4 matmul(const M4& l, const M4& r) {
    M4 out;
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row) {
            float acc = l.at(row, 0) * r.at(0, c);
            for (int k = 1; k < 4; ++k) acc = l.at(row, k) * r.at(k, c) + acc;
            out.at(row, c) = acc;
        }
    return out;
}

// 3x3 minor-based cofactor c(i, j) exactly as the unrolled 4x4 inverse of nalgebra / MESA writes it
M4 invert(const M4& in) {
    const float* m = in.a;
    float v[16];
    v[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    v[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    v[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    v[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    float det = m[0] * v[0] + m[1] * v[4] + m[2] * v[8] + m[3] * v[12];
    M4 out{};
    if (det == 0.0f) return out;          // glm::inverse of a singular matrix is all zeros
    v[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    v[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    v[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    v[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    v[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    v[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    v[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    v[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    v[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    v[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    v[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    v[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float inv_det = 1.0f / det;
    for (int i = 0; i < 16; ++i) out.a[i] = v[i] * inv_det;
    return out;
}

float to_radians(float degrees) { return degrees * (3.14159265358979323846f / 180.0f); }

// m * Rotation(angle about basis axis k): glm::rotate_x/y/z.  nalgebra normalises (axis * angle)
// into a unit axis and a non-negative angle, then fills in Rodrigues' formula.
M4 rotated(const M4& m, float angle, int k) {
    float s[3] = {0.0f * angle, 0.0f * angle, 0.0f * angle};
    s[k] = 1.0f * angle;
    float len = std::sqrt((s[0] * s[0] + s[1] * s[1]) + s[2] * s[2]);
    M4 rot = identity();
    if (len != 0.0f) {
        float x = s[0] / len, y = s[1] / len, z = s[2] / len;
        float xx = x * x, yy = y * y, zz = z * z;
        float sn = std::sin(len), cs = std::cos(len), t = 1.0f - cs;
        rot.at(0, 0) = xx + (1.0f - xx) * cs;  rot.at(0, 1) = x * y * t - z * sn;     rot.at(0, 2) = x * z * t + y * sn;
        rot.at(1, 0) = x * y * t + z * sn;     rot.at(1, 1) = yy + (1.0f - yy) * cs;  rot.at(1, 2) = y * z * t - x * sn;
        rot.at(2, 0) = x * z * t - y * sn;     rot.at(2, 1) = y * z * t + x * sn;     rot.at(2, 2) = zz + (1.0f - zz) * cs;
    }
    return matmul(m, rot);
}

// transform.rs:14-21
M4 rotation(float yaw, float pitch, float roll) { return rotated(rotated(rotated(identity(), yaw, 1), pitch, 0), roll, 2); }

M4 translation(const float p[3]) { M4 m = identity(); m.at(0, 3) = p[0]; m.at(1, 3) = p[1]; m.at(2, 3) = p[2]; return m; }
M4 scaling(const float s[3]) { M4 m = identity(); m.at(0, 0) = s[0]; m.at(1, 1) = s[1]; m.at(2, 2) = s[2]; return m; }

struct CameraState {
    float position[3] = {0, 0, 0};
    float orientation[3] = {0, 0, 0};      // radians; x pitch, y yaw, z roll
    uint32_t width = 0, height = 0;
    float y_fov_degrees = 0, z_near = 0, z_far = 0;
};

// transform.rs:65-78 (viewport :33-43, projection :45-51 = nalgebra Perspective3, view :53-63)
M4 screen_to_world(const CameraState& c) {
    float w = (float)c.width, h = (float)c.height;
    float aspect = w / h;
    M4 view = invert(matmul(translation(c.position), rotation(c.orientation[1], c.orientation[0], c.orientation[2])));
    M4 proj = identity();
    float f = 1.0f / std::tan(to_radians(c.y_fov_degrees) / 2.0f);
    proj.at(1, 1) = f;
    proj.at(0, 0) = f / aspect;
    proj.at(2, 2) = (c.z_far + c.z_near) / (c.z_near - c.z_far);
    proj.at(2, 3) = c.z_far * c.z_near * 2.0f / (c.z_near - c.z_far);
    proj.at(3, 3) = 0.0f;
    proj.at(3, 2) = -1.0f;
    M4 vp{};
    vp.at(0, 0) = w / 2.0f;
    vp.at(1, 1) = h / 2.0f;
    vp.at(2, 2) = (c.z_far - c.z_near) / 2.0f;
    vp.at(0, 3) = 0.0f + w / 2.0f;
    vp.at(1, 3) = 0.0f + h / 2.0f;
    vp.at(2, 3) = (c.z_far + c.z_near) / 2.0f;
    vp.at(3, 3) = 1.0f;
    return invert(matmul(matmul(vp, proj), view));
}

// ------------------------------------------------------------------------------------------------
// geometry records
// ------------------------------------------------------------------------------------------------
void normalize_into(const float v[3], float out[3]) {
    float n = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    for (int i = 0; i < 3; ++i) out[i] = v[i] / n;
}

// Triangle::new, triangle.rs:15-23 and 39-44
RtTriangle triangle_record(const float v[3][3], const float n[3][3], uint32_t material) {
    RtTriangle t{};
    memcpy(t.vertices, v, sizeof(t.vertices));
    memcpy(t.normals, n, sizeof(t.normals));
    float ac[3], ab[3], cr[3];
    for (int i = 0; i < 3; ++i) { ac[i] = v[2][i] - v[0][i]; ab[i] = v[1][i] - v[0][i]; }
    cr[0] = ac[1] * ab[2] - ac[2] * ab[1];
    cr[1] = ac[2] * ab[0] - ac[0] * ab[2];
    cr[2] = ac[0] * ab[1] - ac[1] * ab[0];
    normalize_into(cr, t.face_normal);
    t.material = material;
    return t;
}

// Instance::new, instance.rs:17-39 (model = T * S * R, transform.rs:23-31)
RtMeshInstance instance_record(uint32_t mesh, const float pos[3], const float orient[3], const float scale[3], int32_t override_) {
    M4 r = rotation(orient[1], orient[0], orient[2]);
    M4 rs = matmul(scaling(scale), r);
    M4 model = matmul(matmul(translation(pos), scaling(scale)), r);
    M4 rs_inv = invert(rs), model_inv = invert(model);
    RtMeshInstance out{};
    memcpy(out.rotation_scale, rs.a, 64);
    memcpy(out.rotation_scale_inverse, rs_inv.a, 64);
    memcpy(out.model, model.a, 64);
    memcpy(out.model_inverse, model_inv.a, 64);
    out.mesh = mesh;
    out.material_override = override_;
    return out;
}

// ------------------------------------------------------------------------------------------------
// BVH build: bvh/builder.rs
// ------------------------------------------------------------------------------------------------
struct Box {
    float lo[3], hi[3];
    bool any = false;
    void add(const float p[3]) {
        if (!any) { memcpy(lo, p, 12); memcpy(hi, p, 12); any = true; return; }
        for (int i = 0; i < 3; ++i) { lo[i] = std::fmin(lo[i], p[i]); hi[i] = std::fmax(hi[i], p[i]); }
    }
    void merge(const Box& o) { if (o.any) { add(o.lo); add(o.hi); } }
    // aabb.rs:9-19
    float area() const {
        float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        float xy = ex * ey, yz = ey * ez, xz = ex * ez;
        return 2.0f * xy + 2.0f * yz + 2.0f * xz;
    }
};

struct BvhBuilder {
    const std::vector<RtTriangle>& tris;
    std::vector<RtBvhNode>& nodes;
    std::vector<uint32_t> order;        // triangle indices, partitioned in place
    std::vector<uint32_t> scratch;
    std::vector<uint8_t> bucket;        // per position in `order`
    uint32_t max_depth = 0;

    // triangle.rs:29-36 for one axis
    float centroid(uint32_t tri, int axis) const {
        const RtTriangle& t = tris[tri];
        float sum = 0.0f;
        sum = sum + t.vertices[0][axis];
        sum = sum + t.vertices[1][axis];
        sum = sum + t.vertices[2][axis];
        return sum * (1.0f / 3.0f);
    }

    struct Job { uint32_t parent; bool right; uint32_t begin, end, depth; };

    // builder.rs:178-218 for order[begin, end); returns the new node's index and queues its children
    uint32_t emit(const Job& j, std::vector<Job>& jobs) {
        if (j.depth > max_depth) max_depth = j.depth;
        uint32_t n = j.end - j.begin;
        if (n == 0) raise(RT_ERR_INVALID_ARGUMENT, "Tried to build leaf node with 0 primitives");
        Box hull;
        for (uint32_t i = j.begin; i < j.end; ++i)
            for (int k = 0; k < 3; ++k) hull.add(tris[order[i]].vertices[k]);
        RtBvhNode node{};
        memcpy(node.aabb_min, hull.lo, 12);
        memcpy(node.aabb_max, hull.hi, 12);
        uint32_t index = (uint32_t)nodes.size();
        if (n <= RT_LEAF_TRIANGLES) {
            node.is_leaf = 1;
            node.triangle_count = n;
            for (uint32_t i = 0; i < n; ++i) node.triangle_indices[i] = order[j.begin + i];
            nodes.push_back(node);
            return index;
        }
        // builder.rs:100-176: split axis = largest extent of the centroid hull, five bins over it
        Box cb;
        for (uint32_t i = j.begin; i < j.end; ++i) {
            float c[3] = {centroid(order[i], 0), centroid(order[i], 1), centroid(order[i], 2)};
            cb.add(c);
        }
        float ext[3] = {cb.hi[0] - cb.lo[0], cb.hi[1] - cb.lo[1], cb.hi[2] - cb.lo[2]};
        float widest = std::fmax(std::fmax(ext[0], ext[1]), ext[2]);
        int axis = widest == ext[0] ? 0 : (widest == ext[1] ? 1 : 2);
        float begin_on_axis = cb.lo[axis];
        float bin_width = ext[axis] / 5.0f;
        // bucket b holds the triangles with b <= bin id < b+1 (b = 4: everything from 4 up); a NaN
        // bin id (zero-width bins) is in no bucket, as it passes neither of the reference's filters
        Box bucket_hull[5];
        uint32_t bucket_count[5] = {0, 0, 0, 0, 0};
        for (uint32_t i = j.begin; i < j.end; ++i) {
            float id = (centroid(order[i], axis) - begin_on_axis) / bin_width;
            uint8_t b = 5;
            if (id == id) { b = 0; for (int k = 1; k < 5; ++k) if (id >= (float)k) b = (uint8_t)k; }
            bucket[i] = b;
            if (b < 5) {
                ++bucket_count[b];
                for (int k = 0; k < 3; ++k) bucket_hull[b].add(tris[order[i]].vertices[k]);
            }
        }
        float best_cost = std::numeric_limits<float>::infinity();
        int best_split = -1;
        float parent_area = hull.area();
        for (int split = 1; split < 5; ++split) {
            Box l, r;
            uint32_t nl = 0, nr = 0;
            for (int b = 0; b < split; ++b) { l.merge(bucket_hull[b]); nl += bucket_count[b]; }
            for (int b = split; b < 5; ++b) { r.merge(bucket_hull[b]); nr += bucket_count[b]; }
            float p_left = l.any ? l.area() / parent_area : 0.0f;
            float p_right = r.any ? r.area() / parent_area : 0.0f;
            float cost = p_left * (float)nl + p_right * (float)nr;
            if (cost < best_cost) { best_cost = cost; best_split = split; }
        }
        if (best_split < 0) raise(RT_ERR_INVALID_ARGUMENT, "BVH creation failed: All split costs were infinite");
        // stable partition: left part keeps the order of its members, then the right part
        uint32_t nl = 0, nr = 0;
        for (uint32_t i = j.begin; i < j.end; ++i) {
            if (bucket[i] == 5) continue;
            if (bucket[i] < best_split) scratch[j.begin + nl++] = order[i];
        }
        for (uint32_t i = j.begin; i < j.end; ++i) {
            if (bucket[i] == 5) continue;
            if (bucket[i] >= best_split) { scratch[j.begin + nl + nr] = order[i]; ++nr; }
        }
        if (nl == 0 || nr == 0)
            raise(RT_ERR_INVALID_ARGUMENT, "BVH builder: Aborting infinite recursion - left or right child of node is empty");
        std::copy(scratch.begin() + j.begin, scratch.begin() + j.begin + nl + nr, order.begin() + j.begin);
        node.is_leaf = 0;
        node.child_left = node.child_right = 0xFFFFFFFFu;
        nodes.push_back(node);
        jobs.push_back(Job{index, false, j.begin, j.begin + nl, j.depth + 1});
        jobs.push_back(Job{index, true, j.begin + nl, j.begin + nl + nr, j.depth + 1});     // popped first
        return index;
    }
};

// builder.rs:32-68
RtMesh build_bvh(uint32_t tri_begin, uint32_t tri_end, const std::vector<RtTriangle>& tris, std::vector<RtBvhNode>& nodes) {
    BvhBuilder b{tris, nodes, {}, {}, {}, 0};
    uint32_t n = tri_end - tri_begin;
    b.order.resize(n);
    for (uint32_t i = 0; i < n; ++i) b.order[i] = tri_begin + i;
    b.scratch.resize(n);
    b.bucket.resize(n);
    RtMesh mesh{};
    mesh.triangle_begin = tri_begin;
    mesh.triangle_end = tri_end;
    mesh.node_begin = (uint32_t)nodes.size();
    std::vector<BvhBuilder::Job> jobs;
    b.emit(BvhBuilder::Job{0xFFFFFFFFu, false, 0, n, 0}, jobs);
    while (!jobs.empty()) {
        BvhBuilder::Job j = jobs.back();
        jobs.pop_back();
        uint32_t child = b.emit(j, jobs);
        if (j.right) nodes[j.parent].child_right = child; else nodes[j.parent].child_left = child;
    }
    mesh.node_end = (uint32_t)nodes.size();
    mesh.max_depth = b.max_depth;
    return mesh;
}

// ------------------------------------------------------------------------------------------------
// OBJ: the records object_file.rs / tobj's triangulate + single_index path consume
// ------------------------------------------------------------------------------------------------
struct ObjCursor {
    const char* p;
    const char* end;
    void skip_blank() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p; }
    void skip_line() { while (p < end && *p != '\n') ++p; if (p < end) ++p; }
    bool at_eol() const { return p >= end || *p == '\n'; }
    float number() {
        skip_blank();
        const char* q = p;
        while (p < end && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') ++p;
        return strtof(std::string(q, (size_t)(p - q)).c_str(), nullptr);
    }
    // An optionally signed decimal integer at p, never reading at or beyond `end` (the buffer need not be
    // NUL-terminated) nor across white space (strtol would skip a line break and take the next line's number).
    // Returns false, leaving p alone, when there is no digit.
    bool integer(long& out) {
        const char* q = p;
        bool negative = false;
        if (q < end && (*q == '-' || *q == '+')) { negative = *q == '-'; ++q; }
        if (q >= end || *q < '0' || *q > '9') return false;
        long v = 0;
        while (q < end && *q >= '0' && *q <= '9') { if (v < (1L << 40)) v = v * 10 + (*q - '0'); ++q; }
        out = negative ? -v : v;
        p = q;
        return true;
    }
    // one "v", "v/vt", "v//vn" or "v/vt/vn" corner; returns false at end of line
    bool corner(long& v, long& vn, bool& has_vn) {
        skip_blank();
        if (at_eol()) return false;
        if (!integer(v)) { v = 0; while (p < end && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n' && *p != '/') ++p; }   // strtol gives 0 for no digits
        has_vn = false;
        if (p < end && *p == '/') {
            ++p;
            long vt;
            (void)integer(vt);
            if (p < end && *p == '/') {
                ++p;
                has_vn = integer(vn);
            }
        }
        // whatever else clings to the corner is skipped, so that the caller sees the next corner or the end of the line
        while (p < end && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') ++p;
        return true;
    }
};

// object_file.rs:18-61
RtMesh load_obj(const std::string& name, const char* data, size_t len, uint32_t material, bool counter_clockwise,
                std::vector<RtTriangle>& mesh_triangles, std::vector<RtBvhNode>& nodes) {
    std::vector<float> pos, nrm;
    std::vector<long> corners;                 // (v, vn) pairs, three corners per triangle
    ObjCursor c{data, data + len};
    auto bad = [&](const char* why) { raise(RT_ERR_PARSE, "Failed to load .obj buffer of " + name + ": " + why); };
    while (c.p < c.end) {
        c.skip_blank();
        if (c.end - c.p >= 2 && c.p[0] == 'v' && (c.p[1] == ' ' || c.p[1] == '\t')) {
            c.p += 1;
            for (int i = 0; i < 3; ++i) pos.push_back(c.number());
        } else if (c.end - c.p >= 3 && c.p[0] == 'v' && c.p[1] == 'n' && (c.p[2] == ' ' || c.p[2] == '\t')) {
            c.p += 2;
            for (int i = 0; i < 3; ++i) nrm.push_back(c.number());
        } else if (c.end - c.p >= 2 && c.p[0] == 'f' && (c.p[1] == ' ' || c.p[1] == '\t')) {
            c.p += 1;
            long first[2] = {0, 0}, prev[2] = {0, 0};
            int k = 0;
            long v = 0, vn = 0;
            bool has_vn = false;
            while (c.corner(v, vn, has_vn)) {
                if (!has_vn) bad("Mesh doesn't have exactly one normal per vertex");
                long vi = v > 0 ? v - 1 : (long)(pos.size() / 3) + v;
                long ni = vn > 0 ? vn - 1 : (long)(nrm.size() / 3) + vn;
                if (vi < 0 || vi >= (long)(pos.size() / 3) || ni < 0 || ni >= (long)(nrm.size() / 3)) bad("face index out of range");
                if (k == 0) { first[0] = vi; first[1] = ni; }
                else if (k >= 2) {                       // fan (0, k-1, k)
                    corners.insert(corners.end(), {first[0], first[1], prev[0], prev[1], vi, ni});
                }
                prev[0] = vi; prev[1] = ni;
                ++k;
            }
        }
        c.skip_line();
    }
    uint32_t n_tris = (uint32_t)(corners.size() / 6);
    if (n_tris == 0) raise(RT_ERR_INVALID_ARGUMENT, "Tried to build leaf node with 0 primitives");
    uint32_t begin = (uint32_t)mesh_triangles.size();
    mesh_triangles.reserve(mesh_triangles.size() + n_tris);
    for (uint32_t t = 0; t < n_tris; ++t) {
        float v[3][3], n[3][3];
        for (int k = 0; k < 3; ++k) {
            long vi = corners[6 * (size_t)t + 2 * k], ni = corners[6 * (size_t)t + 2 * k + 1];
            for (int a = 0; a < 3; ++a) {
                v[k][a] = pos[3 * vi + a];
                n[k][a] = counter_clockwise ? -nrm[3 * ni + a] : nrm[3 * ni + a];      // object_file.rs:44-48
            }
        }
        mesh_triangles.push_back(triangle_record(v, n, material));
    }
    return build_bvh(begin, begin + n_tris, mesh_triangles, nodes);
}

// ------------------------------------------------------------------------------------------------
// JSON (the subset serde_json needs for json_format.rs): a flat arena of nodes
// ------------------------------------------------------------------------------------------------
struct Json {
    enum Type { Null, Bool, Num, Str, Arr, Obj };
    struct Node {
        Type type = Null;
        double num = 0;
        std::string str;
        std::vector<int> items;                       // Arr
        std::vector<std::pair<std::string, int>> members;   // Obj
    };
    std::vector<Node> nodes;
    const char* p = nullptr;
    const char* end = nullptr;

    int parse(const char* text, size_t len) {
        p = text; end = text + len;
        int root = value();
        space();
        if (p != end) raise(RT_ERR_PARSE, "scene json: trailing characters");
        return root;
    }
    void space() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    void expect(char c) { space(); if (p >= end || *p != c) raise(RT_ERR_PARSE, std::string("scene json: expected '") + c + "'"); ++p; }
    std::string string_literal() {
        expect('"');
        std::string s;
        while (p < end && *p != '"') {
            char ch = *p++;
            if (ch == '\\' && p < end) {
                char e = *p++;
                ch = e == 'n' ? '\n' : e == 't' ? '\t' : e == 'r' ? '\r' : e;
            }
            s.push_back(ch);
        }
        if (p >= end) raise(RT_ERR_PARSE, "scene json: unterminated string");
        ++p;
        return s;
    }
    int value() {
        space();
        if (p >= end) raise(RT_ERR_PARSE, "scene json: unexpected end of input");
        int id = (int)nodes.size();
        nodes.emplace_back();
        if (*p == '{') {
            ++p;
            nodes[id].type = Obj;
            space();
            if (p < end && *p == '}') { ++p; return id; }
            for (;;) {
                std::string key = string_literal();
                expect(':');
                int child = value();
                nodes[id].members.emplace_back(key, child);
                space();
                if (p < end && *p == ',') { ++p; continue; }
                expect('}');
                return id;
            }
        }
        if (*p == '[') {
            ++p;
            nodes[id].type = Arr;
            space();
            if (p < end && *p == ']') { ++p; return id; }
            for (;;) {
                int child = value();
                nodes[id].items.push_back(child);
                space();
                if (p < end && *p == ',') { ++p; continue; }
                expect(']');
                return id;
            }
        }
        if (*p == '"') { nodes[id].type = Str; std::string s = string_literal(); nodes[id].str = s; return id; }
        auto word = [&](const char* w) { size_t n = strlen(w); if ((size_t)(end - p) >= n && !strncmp(p, w, n)) { p += n; return true; } return false; };
        if (word("true")) { nodes[id].type = Bool; nodes[id].num = 1; return id; }
        if (word("false")) { nodes[id].type = Bool; return id; }
        if (word("null")) return id;
        // a JSON number, copied out first: the text need not be NUL-terminated and strtod must not read beyond `end`
        // (the copy is as long as the token: serde_json takes literals of any length, e.g. 0.1000...0 with a hundred digits)
        size_t n = 0;
        while (p + n < end && (isdigit((unsigned char)p[n]) || p[n] == '-' || p[n] == '+' || p[n] == '.' || p[n] == 'e' || p[n] == 'E')) ++n;
        const std::string token(p, n);
        char* q = nullptr;
        double d = strtod(token.c_str(), &q);
        if (q == token.c_str()) raise(RT_ERR_PARSE, "scene json: unexpected character");
        p += q - token.c_str();
        nodes[id].type = Num;
        nodes[id].num = d;
        return id;
    }
    int member(int obj, const char* key) const {
        if (nodes[obj].type != Obj) return -1;
        for (auto& m : nodes[obj].members) if (m.first == key) return m.second;
        return -1;
    }
    int require(int obj, const char* key) const {
        int id = member(obj, key);
        if (id < 0) raise(RT_ERR_PARSE, std::string("scene json: missing field `") + key + "`");
        return id;
    }
    float f32(int id) const {
        if (nodes[id].type != Num) raise(RT_ERR_PARSE, "scene json: expected a number");
        return (float)nodes[id].num;          // serde_json: f64, then narrowed
    }
    const std::string& text(int id) const {
        if (nodes[id].type != Str) raise(RT_ERR_PARSE, "scene json: expected a string");
        return nodes[id].str;
    }
    void xyz(int id, float out[3]) const { out[0] = f32(require(id, "x")); out[1] = f32(require(id, "y")); out[2] = f32(require(id, "z")); }
    void rgb(int id, float out[3]) const { out[0] = f32(require(id, "r")); out[1] = f32(require(id, "g")); out[2] = f32(require(id, "b")); }
    // Option<Vec<T>>: absent or null -> empty
    std::vector<int> optional_array(int obj, const char* key) const {
        int id = member(obj, key);
        if (id < 0 || nodes[id].type == Null) return {};
        if (nodes[id].type != Arr) raise(RT_ERR_PARSE, std::string("scene json: `") + key + "` is not an array");
        return nodes[id].items;
    }
};

}  // namespace

// ------------------------------------------------------------------------------------------------
// the scene object behind the C ABI
// ------------------------------------------------------------------------------------------------
struct RtHostScene {
    CameraState camera;
    M4 s2w{};
    RtBackground background{};
    std::vector<RtLight> lights;
    std::vector<RtMaterial> materials;
    std::vector<std::string> material_names;
    std::vector<RtPlane> planes;
    std::vector<RtSphere> spheres;
    std::vector<RtTriangle> triangles;
    std::vector<RtTriangle> mesh_triangles;
    std::vector<RtBvhNode> nodes;
    std::vector<RtMesh> meshes;
    std::vector<std::string> mesh_names;
    std::vector<RtMeshInstance> instances;

    int material_index(const std::string& name) const {
        for (size_t i = 0; i < material_names.size(); ++i) if (material_names[i] == name) return (int)i;
        return -1;
    }
    int mesh_index(const std::string& name) const {
        for (size_t i = 0; i < mesh_names.size(); ++i) if (mesh_names[i] == name) return (int)i;
        return -1;
    }
    uint32_t material_or_raise(const std::string& name) const {
        int i = material_index(name);
        if (i < 0) raise(RT_ERR_PARSE, "scene json: unknown material `" + name + "`");     // parser.rs:151 unwrap
        return (uint32_t)i;
    }
};

namespace {

void parse_into(RtHostScene& s, const char* text, size_t len, RtMeshLoadFn loader, void* user) {
    Json j;
    int root = j.parse(text, len);

    // parser.rs:35-47
    int cam = j.require(root, "camera");
    j.xyz(j.require(cam, "position"), s.camera.position);
    float deg[3];
    j.xyz(j.require(cam, "orientation_degrees"), deg);
    for (int i = 0; i < 3; ++i) s.camera.orientation[i] = to_radians(deg[i]);
    int dims = j.require(cam, "screen_dimensions");
    if (j.nodes[dims].type != Json::Arr || j.nodes[dims].items.size() != 2) raise(RT_ERR_PARSE, "scene json: screen_dimensions must hold two integers");
    s.camera.width = (uint32_t)j.nodes[j.nodes[dims].items[0]].num;
    s.camera.height = (uint32_t)j.nodes[j.nodes[dims].items[1]].num;
    s.camera.y_fov_degrees = j.f32(j.require(cam, "y_fov_degrees"));
    s.camera.z_near = j.f32(j.require(cam, "z_near"));
    s.camera.z_far = j.f32(j.require(cam, "z_far"));
    s.s2w = screen_to_world(s.camera);

    // parser.rs:49-58 (externally tagged enum)
    int bg = j.require(root, "background");
    if (j.nodes[bg].type == Json::Str && j.nodes[bg].str == "ColoredDirection") {
        s.background.type = RT_BACKGROUND_COLORED_DIRECTION;
    } else if (j.member(bg, "SolidColor") >= 0) {
        s.background.type = RT_BACKGROUND_SOLID_COLOR;
        j.rgb(j.member(bg, "SolidColor"), s.background.color);
    } else {
        raise(RT_ERR_PARSE, "scene json: unknown background variant");
    }

    int lights = j.require(root, "lights");
    if (j.nodes[lights].type != Json::Arr) raise(RT_ERR_PARSE, "scene json: `lights` is not an array");
    for (int id : j.nodes[lights].items) {                                   // parser.rs:88-99
        RtLight l{};
        int pos = j.require(id, "position");
        j.xyz(pos, l.position);
        l.position[3] = j.f32(j.require(pos, "w"));
        int col = j.require(id, "color");
        j.rgb(j.require(col, "ambient"), l.ambient);
        j.rgb(j.require(col, "diffuse"), l.diffuse);
        j.rgb(j.require(col, "specular"), l.specular);
        s.lights.push_back(l);
    }
    int mats = j.require(root, "materials");
    if (j.nodes[mats].type != Json::Arr) raise(RT_ERR_PARSE, "scene json: `materials` is not an array");
    for (int id : j.nodes[mats].items) {                                     // parser.rs:101-122
        RtMaterial m{};
        j.rgb(j.require(id, "emissive"), m.emissive);
        j.rgb(j.require(id, "ambient"), m.ambient);
        j.rgb(j.require(id, "diffuse"), m.diffuse);
        j.rgb(j.require(id, "specular"), m.specular);
        m.shininess = j.f32(j.require(id, "shininess"));
        int mt = j.require(id, "material_type");
        if (j.nodes[mt].type == Json::Str && j.nodes[mt].str == "Phong") m.material_type = RT_MATERIAL_PHONG;
        else if (j.nodes[mt].type == Json::Str && j.nodes[mt].str == "ReflectAndPhong") m.material_type = RT_MATERIAL_REFLECT_AND_PHONG;
        else if (j.member(mt, "ReflectAndRefract") >= 0) {
            int rr = j.member(mt, "ReflectAndRefract");
            m.material_type = RT_MATERIAL_REFLECT_AND_REFRACT;
            m.index_inner = j.f32(j.require(rr, "index_inner"));
            m.index_outer = j.f32(j.require(rr, "index_outer"));
        } else raise(RT_ERR_PARSE, "scene json: unknown material_type variant");
        s.materials.push_back(m);
        s.material_names.push_back(j.text(j.require(id, "name")));
    }
    for (int id : j.optional_array(root, "planes")) {                        // parser.rs:146-156, plane.rs:15-23
        RtPlane p{};
        float n[3];
        j.xyz(j.require(id, "normal"), n);
        normalize_into(n, p.normal);
        p.distance = j.f32(j.require(id, "distance"));
        p.material = s.material_or_raise(j.text(j.require(id, "material")));
        s.planes.push_back(p);
    }
    for (int id : j.optional_array(root, "spheres")) {                       // parser.rs:158-171
        RtSphere sp{};
        j.xyz(j.require(id, "center"), sp.center);
        sp.radius = j.f32(j.require(id, "radius"));
        sp.material = s.material_or_raise(j.text(j.require(id, "material")));
        s.spheres.push_back(sp);
    }
    for (int id : j.optional_array(root, "triangles")) {                     // parser.rs:173-186
        int vs = j.require(id, "vertices"), ns = j.require(id, "normals");
        if (j.nodes[vs].items.size() != 3 || j.nodes[ns].items.size() != 3) raise(RT_ERR_PARSE, "scene json: a triangle has three vertices and three normals");
        float v[3][3], n[3][3];
        for (int k = 0; k < 3; ++k) { j.xyz(j.nodes[vs].items[k], v[k]); j.xyz(j.nodes[ns].items[k], n[k]); }
        s.triangles.push_back(triangle_record(v, n, s.material_or_raise(j.text(j.require(id, "material")))));
    }
    for (int id : j.optional_array(root, "meshes")) {                        // parser.rs:188-209
        uint32_t material = s.material_or_raise(j.text(j.require(id, "material")));
        const std::string& winding = j.text(j.require(id, "winding_order"));
        if (winding != "Clockwise" && winding != "CounterClockwise") raise(RT_ERR_PARSE, "scene json: unknown winding_order");
        const std::string& file_name = j.text(j.require(id, "file_name"));
        const std::string& name = j.text(j.require(id, "name"));
        if (!loader) raise(RT_ERR_IO, "scene has meshes but no mesh loader was given");
        const char* data = nullptr;
        size_t n = 0;
        void (*release)(void*, const char*, size_t) = nullptr;
        if (loader(user, file_name.c_str(), &data, &n, &release) != 0 || !data)
            raise(RT_ERR_IO, "mesh loader failed for `" + file_name + "`");
        try {
            s.meshes.push_back(load_obj(name, data, n, material, winding == "CounterClockwise", s.mesh_triangles, s.nodes));
        } catch (...) {
            if (release) release(user, data, n);
            throw;
        }
        if (release) release(user, data, n);
        s.mesh_names.push_back(name);
    }
    for (int id : j.optional_array(root, "mesh_instances")) {                // parser.rs:211-230
        int mesh = s.mesh_index(j.text(j.require(id, "mesh")));
        if (mesh < 0) raise(RT_ERR_PARSE, "scene json: unknown mesh `" + j.text(j.require(id, "mesh")) + "`");
        int ov = j.member(id, "material_override");
        int32_t override_ = -1;
        if (ov >= 0 && j.nodes[ov].type == Json::Str) override_ = (int32_t)s.material_or_raise(j.nodes[ov].str);
        float pos[3], deg2[3], scale[3], orient[3];
        j.xyz(j.require(id, "position"), pos);
        j.xyz(j.require(id, "orientation_degrees"), deg2);
        j.xyz(j.require(id, "scale"), scale);
        for (int i = 0; i < 3; ++i) orient[i] = to_radians(deg2[i]);
        s.instances.push_back(instance_record((uint32_t)mesh, pos, orient, scale, override_));
    }
}

// FilesystemMeshLoader, native_app/src/filesystem_mesh_loader.rs:11-30
struct DirLoader { std::string dir; };
void dir_release(void*, const char* data, size_t) { delete[] data; }
int dir_load(void* user, const char* file_name, const char** out_data, size_t* out_len,
             void (**out_release)(void*, const char*, size_t)) {
    const DirLoader* d = static_cast<const DirLoader*>(user);
    std::string path = d->dir.empty() ? std::string(file_name) : d->dir + "/" + file_name;
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return 1;
    std::streamsize n = f.tellg();
    f.seekg(0);
    char* buf = new char[(size_t)n + 1];
    if (n > 0 && !f.read(buf, n)) { delete[] buf; return 1; }
    buf[n] = 0;
    *out_data = buf;
    *out_len = (size_t)n;
    *out_release = dir_release;
    return 0;
}

template <class F>
int guarded(F&& body) {
    try {
        body();
        return RT_OK;
    } catch (const Error& e) {
        return fail(e.code, "%s", e.what.c_str());
    } catch (const std::exception& e) {
        return fail(RT_ERR_INVALID_ARGUMENT, "%s", e.what());
    }
}

}  // namespace

extern "C" {

int rt_host_scene_parse(const char* json_text, size_t json_len, RtMeshLoadFn loader, void* loader_user, RtHostScene** out_scene) {
    if (!json_text || !out_scene) return fail(RT_ERR_INVALID_ARGUMENT, "rt_host_scene_parse: null argument");
    *out_scene = nullptr;
    RtHostScene* s = new RtHostScene();
    int rc = guarded([&] { parse_into(*s, json_text, json_len, loader, loader_user); });
    if (rc != RT_OK) { delete s; return rc; }
    *out_scene = s;
    return RT_OK;
}

int rt_host_scene_load(const char* json_path, const char* model_dir, RtHostScene** out_scene) {
    if (!json_path || !out_scene) return fail(RT_ERR_INVALID_ARGUMENT, "rt_host_scene_load: null argument");
    *out_scene = nullptr;
    std::ifstream f(json_path, std::ios::binary);
    if (!f) return fail(RT_ERR_IO, "cannot open %s", json_path);
    std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    DirLoader loader{model_dir ? model_dir : ""};
    return rt_host_scene_parse(text.data(), text.size(), dir_load, &loader, out_scene);
}

int rt_host_scene_desc(const RtHostScene* s, RtSceneDesc* d) {
    if (!s || !d) return fail(RT_ERR_INVALID_ARGUMENT, "rt_host_scene_desc: null argument");
    memset(d, 0, sizeof(*d));
    d->background = s->background;
    d->n_lights = (uint32_t)s->lights.size();                   d->lights = s->lights.data();
    d->n_materials = (uint32_t)s->materials.size();             d->materials = s->materials.data();
    d->n_planes = (uint32_t)s->planes.size();                   d->planes = s->planes.data();
    d->n_spheres = (uint32_t)s->spheres.size();                 d->spheres = s->spheres.data();
    d->n_triangles = (uint32_t)s->triangles.size();             d->triangles = s->triangles.data();
    d->n_mesh_triangles = (uint32_t)s->mesh_triangles.size();   d->mesh_triangles = s->mesh_triangles.data();
    d->n_mesh_bvh_nodes = (uint32_t)s->nodes.size();            d->mesh_bvh_nodes = s->nodes.data();
    d->n_meshes = (uint32_t)s->meshes.size();                   d->meshes = s->meshes.data();
    d->n_mesh_instances = (uint32_t)s->instances.size();        d->mesh_instances = s->instances.data();
    return RT_OK;
}

int rt_host_scene_camera(const RtHostScene* s, float m[16], uint32_t* width, uint32_t* height) {
    if (!s || !m || !width || !height) return fail(RT_ERR_INVALID_ARGUMENT, "rt_host_scene_camera: null argument");
    memcpy(m, s->s2w.a, 64);
    *width = s->camera.width;
    *height = s->camera.height;
    return RT_OK;
}

int rt_host_scene_resize_screen(RtHostScene* s, uint32_t width, uint32_t height) {
    if (!s) return fail(RT_ERR_INVALID_ARGUMENT, "rt_host_scene_resize_screen: null argument");
    s->camera.width = width;
    s->camera.height = height;
    s->s2w = screen_to_world(s->camera);
    return RT_OK;
}

// scene.rs:60-97
int rt_host_scene_turn_camera(RtHostScene* s, float bx, float by, float ex, float ey) {
    if (!s) return fail(RT_ERR_INVALID_ARGUMENT, "rt_host_scene_turn_camera: null argument");
    CameraState& c = s->camera;
    float degrees_per_pixel = c.y_fov_degrees / (float)c.height;
    float heading = to_radians((ex - bx) * degrees_per_pixel);
    float pitch = to_radians((ey - by) * degrees_per_pixel);
    heading = bx < ex ? std::fabs(heading) : -std::fabs(heading);     // natural scrolling
    pitch = by > ey ? std::fabs(pitch) : -std::fabs(pitch);
    c.orientation[0] += pitch;
    c.orientation[1] += heading;
    float lo = to_radians(-90.0f), hi = to_radians(90.0f);
    if (c.orientation[0] < lo) c.orientation[0] = lo;
    if (c.orientation[0] > hi) c.orientation[0] = hi;
    c.orientation[1] = std::fmod(c.orientation[1], to_radians(360.0f));
    s->s2w = screen_to_world(c);
    return RT_OK;
}

int rt_host_scene_set_camera_pose(RtHostScene* s, const float position[3], const float orientation[3]) {
    if (!s) return fail(RT_ERR_INVALID_ARGUMENT, "rt_host_scene_set_camera_pose: null argument");
    if (position) memcpy(s->camera.position, position, 12);
    if (orientation) memcpy(s->camera.orientation, orientation, 12);
    s->s2w = screen_to_world(s->camera);
    return RT_OK;
}

int rt_host_scene_get_camera_pose(const RtHostScene* s, float position[3], float orientation[3]) {
    if (!s) return fail(RT_ERR_INVALID_ARGUMENT, "rt_host_scene_get_camera_pose: null argument");
    if (position) memcpy(position, s->camera.position, 12);
    if (orientation) memcpy(orientation, s->camera.orientation, 12);
    return RT_OK;
}

int rt_host_scene_find_material(const RtHostScene* s, const char* name) { return (s && name) ? s->material_index(name) : -1; }
int rt_host_scene_find_mesh(const RtHostScene* s, const char* name) { return (s && name) ? s->mesh_index(name) : -1; }

int rt_host_scene_add_instance(RtHostScene* s, uint32_t mesh, const float position[3], const float orientation[3],
                               const float scale[3], int32_t material_override) {
    if (!s || !position || !orientation || !scale) return fail(RT_ERR_INVALID_ARGUMENT, "rt_host_scene_add_instance: null argument");
    if (mesh >= s->meshes.size()) return fail(RT_ERR_OUT_OF_RANGE, "mesh %u outside the %zu meshes", mesh, s->meshes.size());
    if (material_override < -1 || material_override >= (int32_t)s->materials.size())
        return fail(RT_ERR_OUT_OF_RANGE, "material_override %d outside the %zu materials", material_override, s->materials.size());
    s->instances.push_back(instance_record(mesh, position, orientation, scale, material_override));
    return RT_OK;
}

int rt_host_scene_add_mesh(RtHostScene* s, const char* name, const char* obj_data, size_t obj_len, uint32_t material,
                           int counter_clockwise) {
    if (!s || !name || !obj_data) return fail(RT_ERR_INVALID_ARGUMENT, "rt_host_scene_add_mesh: null argument");
    if (material >= s->materials.size()) return fail(RT_ERR_OUT_OF_RANGE, "material %u outside the %zu materials", material, s->materials.size());
    size_t tris_before = s->mesh_triangles.size(), nodes_before = s->nodes.size();
    int rc = guarded([&] {
        s->meshes.push_back(load_obj(name, obj_data, obj_len, material, counter_clockwise != 0, s->mesh_triangles, s->nodes));
        s->mesh_names.push_back(name);
    });
    if (rc != RT_OK) { s->mesh_triangles.resize(tris_before); s->nodes.resize(nodes_before); }
    return rc;
}

void rt_host_scene_destroy(RtHostScene* s) { delete s; }

}  // extern "C"


// ---- PNG output (native_app/src/write_png.rs) -------------------------------------------------------
namespace {
struct Crc32 {
    uint32_t table[256];
    Crc32() {
        for (uint32_t n = 0; n < 256; ++n) {
            uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[n] = c;
        }
    }
    uint32_t run(uint32_t crc, const uint8_t* p, size_t n) const {
        for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
        return crc;
    }
};

void put_u32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

bool write_chunk(FILE* f, const Crc32& crc, const char type[4], const std::vector<uint8_t>& data) {
    std::vector<uint8_t> head;
    put_u32(head, (uint32_t)data.size());
    head.insert(head.end(), type, type + 4);
    uint32_t c = crc.run(0xFFFFFFFFu, head.data() + 4, 4);
    c = crc.run(c, data.data(), data.size()) ^ 0xFFFFFFFFu;
    std::vector<uint8_t> tail;
    put_u32(tail, c);
    return fwrite(head.data(), 1, head.size(), f) == head.size() &&
           (data.empty() || fwrite(data.data(), 1, data.size(), f) == data.size()) && fwrite(tail.data(), 1, 4, f) == 4;
}
}  // namespace

extern "C" int rt_host_write_png(const char* path, const uint8_t* pixels, uint32_t width, uint32_t height, uint32_t channels) {
    if (!path || !pixels) return fail(RT_ERR_INVALID_ARGUMENT, "rt_host_write_png: null argument");
    if (width == 0 || height == 0 || (channels != 3 && channels != 4))
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_host_write_png: %ux%u with %u channels", width, height, channels);
    const size_t row = (size_t)width * channels;
    // filtered scanlines: filter type 0 (None) in front of every row
    std::vector<uint8_t> raw;
    raw.reserve((row + 1) * height);
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), pixels + y * row, pixels + (y + 1) * row);
    }
    // zlib stream of stored blocks
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t at = 0; at < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - at);
        z.push_back(at + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + at, raw.begin() + at + n);
        for (size_t i = at; i < at + n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        at += n;
    }
    put_u32(z, (b << 16) | a);
    FILE* f = fopen(path, "wb");
    if (!f) return fail(RT_ERR_IO, "cannot create %s", path);
    static const Crc32 crc;
    static const uint8_t signature[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_u32(ihdr, width); put_u32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(channels == 4 ? 6 : 2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    bool ok = fwrite(signature, 1, 8, f) == 8 && write_chunk(f, crc, "IHDR", ihdr) && write_chunk(f, crc, "IDAT", z) &&
              write_chunk(f, crc, "IEND", std::vector<uint8_t>());
    ok = (fclose(f) == 0) && ok;
    if (!ok) return fail(RT_ERR_IO, "writing %s failed", path);
    return RT_OK;
}
