// TEST INFRASTRUCTURE - CPU oracle, not part of the shipped renderer.
//
// Setup-time half of the oracle: everything the reference does between reading a scene file and
// having a lib_raytracer::Scene in memory.  Each function names the reference lines it restates.
#include "orc_scene.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace orc {

// ---------------------------------------------------------------------------------------------
// transform.rs
// ---------------------------------------------------------------------------------------------

// transform.rs:6-8 (glm::translation = identity with the offset in column 3)
Mat4 m_translation(Vec3 o) {
    Mat4 r = mat_identity();
    r.m[12] = o.x; r.m[13] = o.y; r.m[14] = o.z;
    return r;
}

// transform.rs:10-12 (glm::scaling = diagonal)
Mat4 m_scaling(Vec3 s) {
    Mat4 r = mat_identity();
    r.m[0] = s.x; r.m[5] = s.y; r.m[10] = s.z;
    return r;
}

// glm::rotate_{x,y,z}(m, angle) = m * Matrix4::new_rotation(axis * angle).  nalgebra turns the
// scaled axis into (unit axis, angle = its norm) and builds the Rodrigues matrix from those; a
// zero angle gives the identity.  `axis` is 0, 1 or 2.
static Mat4 rotate_axis(const Mat4& m, float angle, int axis) {
    float scaled[3] = {0.0f * angle, 0.0f * angle, 0.0f * angle};
    scaled[axis] = 1.0f * angle;
    float n = std::sqrt((scaled[0] * scaled[0] + scaled[1] * scaled[1]) + scaled[2] * scaled[2]);
    if (!(n != 0.0f)) return mul(m, mat_identity());
    float ux = scaled[0] / n, uy = scaled[1] / n, uz = scaled[2] / n;
    float sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
    float sn = std::sin(n), cs = std::cos(n);
    float omc = 1.0f - cs;
    // row-major entries r[row][col] as nalgebra's Matrix3::new lists them
    float r00 = sqx + (1.0f - sqx) * cs;
    float r01 = ux * uy * omc - uz * sn;
    float r02 = ux * uz * omc + uy * sn;
    float r10 = ux * uy * omc + uz * sn;
    float r11 = sqy + (1.0f - sqy) * cs;
    float r12 = uy * uz * omc - ux * sn;
    float r20 = ux * uz * omc - uy * sn;
    float r21 = uy * uz * omc + ux * sn;
    float r22 = sqz + (1.0f - sqz) * cs;
    Mat4 rot = mat_identity();
    rot.m[0] = r00; rot.m[1] = r10; rot.m[2] = r20;
    rot.m[4] = r01; rot.m[5] = r11; rot.m[6] = r21;
    rot.m[8] = r02; rot.m[9] = r12; rot.m[10] = r22;
    return mul(m, rot);
}

// transform.rs:14-21
Mat4 m_rotation(float yaw, float pitch, float roll) {
    Mat4 mat = mat_identity();
    mat = rotate_axis(mat, yaw, 1);
    mat = rotate_axis(mat, pitch, 0);
    return rotate_axis(mat, roll, 2);
}

// transform.rs:23-31
Mat4 m_model(Vec3 position, Vec3 orientation, Vec3 scale) {
    Mat4 s = m_scaling(scale);
    Mat4 r = m_rotation(orientation.y, orientation.x, orientation.z);
    Mat4 t = m_translation(position);
    return mul(mul(t, s), r);
}

// transform.rs:33-43
Mat4 m_viewport(float x, float y, float w, float h, float z_near, float z_far) {
    Mat4 r{};
    r.m[0] = w / 2.0f;
    r.m[5] = h / 2.0f;
    r.m[10] = (z_far - z_near) / 2.0f;
    r.m[12] = x + w / 2.0f;
    r.m[13] = y + h / 2.0f;
    r.m[14] = (z_far + z_near) / 2.0f;
    r.m[15] = 1.0f;
    return r;
}

// transform.rs:45-51: glm::perspective(aspect, fovy, near, far) = nalgebra Perspective3::new:
// identity; m11 = 1/tan(fovy/2); m00 = m11/aspect; m22 = (far+near)/(near-far);
// m23 = far*near*2/(near-far); m33 = 0; m32 = -1.
Mat4 m_projection(float y_fov_degrees, float aspect, float z_near, float z_far) {
    float fovy = radians(y_fov_degrees);
    Mat4 r = mat_identity();
    float m11 = 1.0f / std::tan(fovy / 2.0f);
    r.m[5] = m11;
    r.m[0] = m11 / aspect;
    r.m[10] = (z_far + z_near) / (z_near - z_far);
    r.m[14] = z_far * z_near * 2.0f / (z_near - z_far);
    r.m[15] = 0.0f;
    r.m[11] = -1.0f;
    return r;
}

// transform.rs:53-63
Mat4 m_view(Vec3 orientation, Vec3 position) {
    Mat4 r = m_rotation(orientation.y, orientation.x, orientation.z);
    Mat4 t = m_translation(position);
    return inverse(mul(t, r));
}

// transform.rs:65-78
Mat4 m_screen_to_world(const Camera& c) {
    float w = (float)c.width, h = (float)c.height;
    float aspect = w / h;
    Mat4 view = m_view(c.orientation, c.position);
    Mat4 proj = m_projection(c.y_fov_degrees, aspect, c.z_near, c.z_far);
    Mat4 vp = m_viewport(0.0f, 0.0f, w, h, c.z_near, c.z_far);
    return inverse(mul(mul(vp, proj), view));
}

// scene.rs:54-58
void resize_screen(Scene& s, uint32_t width, uint32_t height) {
    s.camera.width = width;
    s.camera.height = height;
    s.screen_to_world = m_screen_to_world(s.camera);
}

// scene.rs:60-97
void turn_camera(Scene& s, float bx, float by, float ex, float ey) {
    float degrees_per_pixel = s.camera.y_fov_degrees / (float)s.camera.height;
    float dx = ex - bx, dy = ey - by;
    float heading = radians(dx * degrees_per_pixel);
    float pitch = radians(dy * degrees_per_pixel);
    heading = (bx < ex) ? std::fabs(heading) : -std::fabs(heading);
    pitch = (by > ey) ? std::fabs(pitch) : -std::fabs(pitch);
    Vec3& o = s.camera.orientation;
    o.x += pitch;
    o.y += heading;
    float lo = radians(-90.0f), hi = radians(90.0f);
    // f32::clamp
    if (o.x < lo) o.x = lo;
    if (o.x > hi) o.x = hi;
    o.y = std::fmod(o.y, radians(360.0f));
    s.screen_to_world = m_screen_to_world(s.camera);
}

// ---------------------------------------------------------------------------------------------
// geometry constructors
// ---------------------------------------------------------------------------------------------

static void put3(float* dst, Vec3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

// plane.rs:15-23
RtPlane make_plane(Vec3 normal, float distance, uint32_t material) {
    RtPlane p{};
    put3(p.normal, normalize(normal));
    p.distance = distance;
    p.material = material;
    return p;
}

// triangle.rs:15-23, 39-44: face normal = normalize(cross(c - a, b - a))
RtTriangle make_triangle(const Vec3 v[3], const Vec3 n[3], uint32_t material) {
    RtTriangle t{};
    for (int i = 0; i < 3; ++i) { put3(t.vertices[i], v[i]); put3(t.normals[i], n[i]); }
    Vec3 ac = v[2] - v[0];
    Vec3 ab = v[1] - v[0];
    put3(t.face_normal, normalize(cross(ac, ab)));
    t.material = material;
    return t;
}

// triangle.rs:29-36
Vec3 triangle_centroid(const RtTriangle& t) {
    Vec3 sum{0.0f, 0.0f, 0.0f};
    for (int i = 0; i < 3; ++i) sum = sum + v3(t.vertices[i]);
    return sum * (1.0f / 3.0f);
}

// instance.rs:17-39
RtMeshInstance make_instance(uint32_t mesh, Vec3 position, Vec3 orientation, Vec3 scale, int32_t material_override) {
    RtMeshInstance inst{};
    Mat4 rs = mul(m_scaling(scale), m_rotation(orientation.y, orientation.x, orientation.z));
    Mat4 rs_inv = inverse(rs);
    Mat4 model = m_model(position, orientation, scale);
    Mat4 model_inv = inverse(model);
    std::memcpy(inst.rotation_scale, rs.m, 64);
    std::memcpy(inst.rotation_scale_inverse, rs_inv.m, 64);
    std::memcpy(inst.model, model.m, 64);
    std::memcpy(inst.model_inverse, model_inv.m, 64);
    inst.mesh = mesh;
    inst.material_override = material_override;
    return inst;
}

// ---------------------------------------------------------------------------------------------
// hull.rs / aabb.rs
// ---------------------------------------------------------------------------------------------

static inline void hull_add(Aabb& r, Vec3 v) {
    r.min.x = fmin32(r.min.x, v.x); r.min.y = fmin32(r.min.y, v.y); r.min.z = fmin32(r.min.z, v.z);
    r.max.x = fmax32(r.max.x, v.x); r.max.y = fmax32(r.max.y, v.y); r.max.z = fmax32(r.max.z, v.z);
}

// hull.rs:17-35
bool hull_from_vertices(const Vec3* vertices, size_t n, Aabb* out) {
    if (n == 0) return false;
    Aabb r{vertices[0], vertices[0]};
    for (size_t i = 1; i < n; ++i) hull_add(r, vertices[i]);
    *out = r;
    return true;
}

// aabb.rs:9-19
float aabb_surface_area(const Aabb& b) {
    Vec3 e = b.max - b.min;
    float xy = e.x * e.y, yz = e.y * e.z, xz = e.x * e.z;
    return 2.0f * xy + 2.0f * yz + 2.0f * xz;
}

// ---------------------------------------------------------------------------------------------
// bvh/builder.rs
// ---------------------------------------------------------------------------------------------

namespace {

struct BuildJob {
    uint32_t parent_index;
    bool right;
    std::vector<uint32_t> triangle_indices;
    uint32_t depth;
};

struct Builder {
    const std::vector<RtTriangle>& tris;
    std::vector<RtBvhNode>& nodes;
    std::vector<BuildJob> job_stack;
    uint32_t max_depth = 0;

    // hull.rs:57-61 over the triangles named by `idx` (each triangle's three vertices in order)
    bool hull_of(const std::vector<uint32_t>& idx, Aabb* out) const {
        if (idx.empty()) return false;
        Aabb r{v3(tris[idx[0]].vertices[0]), v3(tris[idx[0]].vertices[0])};
        bool first = true;
        for (uint32_t i : idx) {
            for (int k = 0; k < 3; ++k) {
                if (first) { first = false; continue; }
                hull_add(r, v3(tris[i].vertices[k]));
            }
        }
        *out = r;
        return true;
    }

    // builder.rs:100-176
    void split(const std::vector<uint32_t>& idx, const Aabb& volume, std::vector<uint32_t>& left, std::vector<uint32_t>& right) const {
        std::vector<Vec3> centroids(idx.size());
        for (size_t i = 0; i < idx.size(); ++i) centroids[i] = triangle_centroid(tris[idx[i]]);
        Aabb cb;
        if (!hull_from_vertices(centroids.data(), centroids.size(), &cb)) throw std::runtime_error("Empty triangles iterator");
        Vec3 extent = cb.max - cb.min;
        float emax = fmax32(fmax32(extent.x, extent.y), extent.z);
        int axis = (emax == extent.x) ? 0 : (emax == extent.y) ? 1 : 2;
        auto get_axis = [axis](Vec3 p) { return axis == 0 ? p.x : axis == 1 ? p.y : p.z; };
        float axis_begin = get_axis(cb.min);
        float axis_extent = get_axis(extent);
        float bin_width = axis_extent / 5.0f;   // BVH::BINS_PER_LAYER, bvh/mod.rs:18
        std::vector<float> bin_id(idx.size());
        for (size_t i = 0; i < idx.size(); ++i) bin_id[i] = (get_axis(centroids[i]) - axis_begin) / bin_width;

        float min_cost = std::numeric_limits<float>::infinity();
        int min_bin = -1;
        std::vector<uint32_t> l, r;
        for (int split_bin = 1; split_bin < 5; ++split_bin) {
            l.clear(); r.clear();
            for (size_t i = 0; i < idx.size(); ++i) {
                if (bin_id[i] < (float)split_bin) l.push_back(idx[i]);
                if (bin_id[i] >= (float)split_bin) r.push_back(idx[i]);
            }
            Aabb lh, rh;
            float p_left = hull_of(l, &lh) ? aabb_surface_area(lh) / aabb_surface_area(volume) : 0.0f;
            float p_right = hull_of(r, &rh) ? aabb_surface_area(rh) / aabb_surface_area(volume) : 0.0f;
            float cost = p_left * (float)l.size() + p_right * (float)r.size();
            if (cost < min_cost) { min_cost = cost; min_bin = split_bin; }
        }
        if (min_bin < 0) throw std::runtime_error("BVH creation failed: All split costs were infinite");
        left.clear(); right.clear();
        for (size_t i = 0; i < idx.size(); ++i) {
            if (bin_id[i] < (float)min_bin) left.push_back(idx[i]);
            if (bin_id[i] >= (float)min_bin) right.push_back(idx[i]);
        }
    }

    // builder.rs:178-218
    uint32_t build_node_queue_children(std::vector<uint32_t> idx, uint32_t depth) {
        if (depth > max_depth) max_depth = depth;
        Aabb aabb;
        if (!hull_of(idx, &aabb)) throw std::runtime_error("Tried to build leaf node with 0 primitives");
        RtBvhNode node{};
        put3(node.aabb_min, aabb.min);
        put3(node.aabb_max, aabb.max);
        uint32_t node_index = (uint32_t)nodes.size();
        if (idx.size() <= RT_LEAF_TRIANGLES) {
            node.is_leaf = 1;
            node.triangle_count = (uint32_t)idx.size();
            for (size_t i = 0; i < idx.size(); ++i) node.triangle_indices[i] = idx[i];
        } else {
            std::vector<uint32_t> left, right;
            split(idx, aabb, left, right);
            if (left.empty() || right.empty())
                throw std::runtime_error("BVH builder: Aborting infinite recursion - left or right child of node is empty");
            node.is_leaf = 0;
            node.child_left = node.child_right = 0xFFFFFFFFu;
            job_stack.push_back(BuildJob{node_index, false, std::move(left), depth + 1});
            job_stack.push_back(BuildJob{node_index, true, std::move(right), depth + 1});
        }
        nodes.push_back(node);
        return node_index;
    }
};

}  // namespace

// builder.rs:32-68
RtMesh build_bvh(uint32_t tri_begin, uint32_t tri_end, const std::vector<RtTriangle>& mesh_triangles,
                 std::vector<RtBvhNode>& mesh_bvh_nodes) {
    uint32_t first = (uint32_t)mesh_bvh_nodes.size();
    Builder b{mesh_triangles, mesh_bvh_nodes, {}, 0};
    std::vector<uint32_t> all;
    for (uint32_t i = tri_begin; i < tri_end; ++i) all.push_back(i);
    uint32_t root = b.build_node_queue_children(std::move(all), 0);
    if (root != first) throw std::runtime_error("BVH root node created in wrong position");
    while (!b.job_stack.empty()) {
        BuildJob job = std::move(b.job_stack.back());
        b.job_stack.pop_back();
        uint32_t child = b.build_node_queue_children(std::move(job.triangle_indices), job.depth);
        RtBvhNode& parent = mesh_bvh_nodes[job.parent_index];
        if (parent.is_leaf) throw std::runtime_error("Tried to attach child to leaf node");
        if (job.right) parent.child_right = child; else parent.child_left = child;
    }
    RtMesh mesh{};
    mesh.triangle_begin = tri_begin;
    mesh.triangle_end = tri_end;
    mesh.node_begin = first;
    mesh.node_end = (uint32_t)mesh_bvh_nodes.size();
    mesh.max_depth = b.max_depth;
    return mesh;
}

// ---------------------------------------------------------------------------------------------
// object_file.rs (with the subset of tobj 4.0.3 it relies on: v / vn / f records, `triangulate`
// = fan (0, i, i+1), 1-based or negative relative indices, `a`, `a/b`, `a//c`, `a/b/c` corners)
// ---------------------------------------------------------------------------------------------

RtMesh load_mesh(const std::string& name, const std::string& obj_text, uint32_t material, bool counter_clockwise,
                 std::vector<RtTriangle>& mesh_triangles, std::vector<RtBvhNode>& mesh_bvh_nodes) {
    std::vector<Vec3> pos, nrm;
    struct Corner { long v, n; };
    std::vector<Corner> corners;          // three per triangle
    const char* p = obj_text.c_str();
    const char* end = p + obj_text.size();
    auto fail = [&](const std::string& why) { throw std::runtime_error("Failed to load .obj buffer of " + name + ": " + why); };
    while (p < end) {
        const char* eol = (const char*)memchr(p, '\n', end - p);
        if (!eol) eol = end;
        std::string line(p, eol);
        p = eol + 1;
        size_t a = line.find_first_not_of(" \t\r");
        if (a == std::string::npos) continue;
        if (line.compare(a, 2, "v ") == 0 || line.compare(a, 2, "v\t") == 0) {
            char* q = nullptr;
            const char* s = line.c_str() + a + 2;
            float x = strtof(s, &q); float y = strtof(q, &q); float z = strtof(q, &q);
            pos.push_back({x, y, z});
        } else if (line.compare(a, 3, "vn ") == 0 || line.compare(a, 3, "vn\t") == 0) {
            char* q = nullptr;
            const char* s = line.c_str() + a + 3;
            float x = strtof(s, &q); float y = strtof(q, &q); float z = strtof(q, &q);
            nrm.push_back({x, y, z});
        } else if (line.compare(a, 2, "f ") == 0 || line.compare(a, 2, "f\t") == 0) {
            std::istringstream ss(line.substr(a + 2));
            std::string tok;
            std::vector<Corner> face;
            while (ss >> tok) {
                long vi = 0, ni = 0;
                bool has_n = false;
                size_t s1 = tok.find('/');
                vi = strtol(tok.substr(0, s1).c_str(), nullptr, 10);
                if (s1 != std::string::npos) {
                    size_t s2 = tok.find('/', s1 + 1);
                    if (s2 != std::string::npos && s2 + 1 < tok.size()) {
                        ni = strtol(tok.substr(s2 + 1).c_str(), nullptr, 10);
                        has_n = true;
                    }
                }
                if (!has_n) fail("Mesh doesn't have exactly one normal per vertex");
                long v = vi > 0 ? vi - 1 : (long)pos.size() + vi;
                long n = ni > 0 ? ni - 1 : (long)nrm.size() + ni;
                if (v < 0 || v >= (long)pos.size() || n < 0 || n >= (long)nrm.size()) fail("face index out of range");
                face.push_back({v, n});
            }
            for (size_t i = 1; i + 1 < face.size(); ++i) {
                corners.push_back(face[0]); corners.push_back(face[i]); corners.push_back(face[i + 1]);
            }
        }
    }
    uint32_t n_tris = (uint32_t)(corners.size() / 3);
    uint32_t begin = n_tris > 0 ? (uint32_t)mesh_triangles.size() : 0;
    uint32_t tri_end = n_tris > 0 ? begin + n_tris : 0;
    for (uint32_t t = 0; t < n_tris; ++t) {
        Vec3 v[3], n[3];
        for (int k = 0; k < 3; ++k) {
            v[k] = pos[corners[3 * t + k].v];
            n[k] = nrm[corners[3 * t + k].n];
            if (counter_clockwise) n[k] = -n[k];          // object_file.rs:44-48
        }
        mesh_triangles.push_back(make_triangle(v, n, material));
    }
    if (n_tris == 0) {
        // BVH::build on an empty range reaches "Tried to build leaf node with 0 primitives"
        throw std::runtime_error("Tried to build leaf node with 0 primitives");
    }
    return build_bvh(begin, tri_end, mesh_triangles, mesh_bvh_nodes);
}

// ---------------------------------------------------------------------------------------------
// minimal JSON reader (serde_json stand-in) + scene_file/parser.rs
// ---------------------------------------------------------------------------------------------

namespace {

struct JValue;
using JPtr = std::shared_ptr<JValue>;
struct JValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<JPtr> arr;
    std::vector<std::pair<std::string, JPtr>> obj;
    const JPtr* find(const std::string& k) const {
        for (auto& kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const char* p;
    const char* end;
    [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("scene json: ") + what); }
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    JPtr value() {
        ws();
        if (p >= end) fail("unexpected end");
        auto v = std::make_shared<JValue>();
        char c = *p;
        if (c == '{') {
            v->kind = JValue::Object; ++p; ws();
            if (p < end && *p == '}') { ++p; return v; }
            for (;;) {
                ws();
                std::string k = string();
                ws();
                if (p >= end || *p != ':') fail("expected ':'");
                ++p;
                v->obj.emplace_back(k, value());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v->kind = JValue::Array; ++p; ws();
            if (p < end && *p == ']') { ++p; return v; }
            for (;;) {
                v->arr.push_back(value());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v->kind = JValue::String; v->str = string();
        } else if (c == 't' && end - p >= 4 && !strncmp(p, "true", 4)) {
            v->kind = JValue::Bool; v->b = true; p += 4;
        } else if (c == 'f' && end - p >= 5 && !strncmp(p, "false", 5)) {
            v->kind = JValue::Bool; v->b = false; p += 5;
        } else if (c == 'n' && end - p >= 4 && !strncmp(p, "null", 4)) {
            v->kind = JValue::Null; p += 4;
        } else {
            char* q = nullptr;
            v->kind = JValue::Number;
            v->num = strtod(p, &q);
            if (q == p) fail("bad number");
            p = q;
        }
        return v;
    }
    std::string string() {
        if (p >= end || *p != '"') fail("expected string");
        ++p;
        std::string s;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                    case 'n': s += '\n'; break; case 't': s += '\t'; break; case 'r': s += '\r'; break;
                    case 'b': s += '\b'; break; case 'f': s += '\f'; break;
                    default: s += *p; break;
                }
                ++p;
            } else {
                s += *p++;
            }
        }
        if (p >= end) fail("unterminated string");
        ++p;
        return s;
    }
};

const JValue& need(const JValue& o, const char* key) {
    const JPtr* v = o.find(key);
    if (!v) throw std::runtime_error(std::string("scene json: missing field `") + key + "`");
    return **v;
}
float f32_of(const JValue& v) {
    if (v.kind != JValue::Number) throw std::runtime_error("scene json: expected a number");
    return (float)v.num;   // serde_json reads the number as f64 and narrows
}
Vec3 vec3_of(const JValue& v) { return {f32_of(need(v, "x")), f32_of(need(v, "y")), f32_of(need(v, "z"))}; }
Vec3 rgb_of(const JValue& v) { return {f32_of(need(v, "r")), f32_of(need(v, "g")), f32_of(need(v, "b"))}; }
const std::vector<JPtr>* opt_array(const JValue& o, const char* key) {
    const JPtr* v = o.find(key);
    if (!v || (*v)->kind == JValue::Null) return nullptr;
    if ((*v)->kind != JValue::Array) throw std::runtime_error(std::string("scene json: `") + key + "` is not an array");
    return &(*v)->arr;
}

// parser.rs:124-133
uint32_t find_material(const Scene& s, const std::string& name) {
    for (size_t i = 0; i < s.material_names.size(); ++i) if (s.material_names[i] == name) return (uint32_t)i;
    throw std::runtime_error("scene json: unknown material `" + name + "`");
}

}  // namespace

std::string read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// parser.rs:61-78 with the conversions of parser.rs:35-58, 88-230
Scene parse_scene_json(const std::string& text, const std::string& model_dir) {
    JParser jp{text.c_str(), text.c_str() + text.size()};
    JPtr rootp = jp.value();
    const JValue& root = *rootp;
    Scene s;

    const JValue& cam = need(root, "camera");
    s.camera.position = vec3_of(need(cam, "position"));
    Vec3 od = vec3_of(need(cam, "orientation_degrees"));
    s.camera.orientation = {radians(od.x), radians(od.y), radians(od.z)};
    const JValue& dims = need(cam, "screen_dimensions");
    if (dims.kind != JValue::Array || dims.arr.size() != 2) throw std::runtime_error("scene json: screen_dimensions");
    s.camera.width = (uint32_t)dims.arr[0]->num;
    s.camera.height = (uint32_t)dims.arr[1]->num;
    s.camera.y_fov_degrees = f32_of(need(cam, "y_fov_degrees"));
    s.camera.z_near = f32_of(need(cam, "z_near"));
    s.camera.z_far = f32_of(need(cam, "z_far"));
    s.screen_to_world = m_screen_to_world(s.camera);   // scene.rs:23-24

    const JValue& bg = need(root, "background");
    if (bg.kind == JValue::String && bg.str == "ColoredDirection") {
        s.background.type = RT_BACKGROUND_COLORED_DIRECTION;
    } else if (bg.kind == JValue::Object && bg.find("SolidColor")) {
        s.background.type = RT_BACKGROUND_SOLID_COLOR;
        put3(s.background.color, rgb_of(need(bg, "SolidColor")));
    } else {
        throw std::runtime_error("scene json: unknown background");
    }

    const JValue& lights = need(root, "lights");
    if (lights.kind != JValue::Array) throw std::runtime_error("scene json: `lights` is not an array");
    for (auto& lp : lights.arr) {
        const JValue& l = *lp;
        RtLight out{};
        const JValue& pos = need(l, "position");
        out.position[0] = f32_of(need(pos, "x")); out.position[1] = f32_of(need(pos, "y"));
        out.position[2] = f32_of(need(pos, "z")); out.position[3] = f32_of(need(pos, "w"));
        const JValue& col = need(l, "color");
        put3(out.ambient, rgb_of(need(col, "ambient")));
        put3(out.diffuse, rgb_of(need(col, "diffuse")));
        put3(out.specular, rgb_of(need(col, "specular")));
        s.lights.push_back(out);
    }

    {
        const JValue& mats = need(root, "materials");
        if (mats.kind != JValue::Array) throw std::runtime_error("scene json: `materials` is not an array");
        for (auto& mp : mats.arr) {
            const JValue& m = *mp;
            RtMaterial out{};
            put3(out.emissive, rgb_of(need(m, "emissive")));
            put3(out.ambient, rgb_of(need(m, "ambient")));
            put3(out.diffuse, rgb_of(need(m, "diffuse")));
            put3(out.specular, rgb_of(need(m, "specular")));
            out.shininess = f32_of(need(m, "shininess"));
            const JValue& mt = need(m, "material_type");
            if (mt.kind == JValue::String && mt.str == "Phong") out.material_type = RT_MATERIAL_PHONG;
            else if (mt.kind == JValue::String && mt.str == "ReflectAndPhong") out.material_type = RT_MATERIAL_REFLECT_AND_PHONG;
            else if (mt.kind == JValue::Object && mt.find("ReflectAndRefract")) {
                const JValue& rr = need(mt, "ReflectAndRefract");
                out.material_type = RT_MATERIAL_REFLECT_AND_REFRACT;
                out.index_inner = f32_of(need(rr, "index_inner"));
                out.index_outer = f32_of(need(rr, "index_outer"));
            } else throw std::runtime_error("scene json: unknown material_type");
            s.materials.push_back(out);
            s.material_names.push_back(need(m, "name").str);
        }
    }

    if (auto* a = opt_array(root, "planes"))
        for (auto& pp : *a)
            s.planes.push_back(make_plane(vec3_of(need(*pp, "normal")), f32_of(need(*pp, "distance")),
                                          find_material(s, need(*pp, "material").str)));
    if (auto* a = opt_array(root, "spheres"))
        for (auto& sp : *a) {
            RtSphere out{};
            put3(out.center, vec3_of(need(*sp, "center")));
            out.radius = f32_of(need(*sp, "radius"));
            out.material = find_material(s, need(*sp, "material").str);
            s.spheres.push_back(out);
        }
    if (auto* a = opt_array(root, "triangles"))
        for (auto& tp : *a) {
            const JValue& vs = need(*tp, "vertices");
            const JValue& ns = need(*tp, "normals");
            if (vs.arr.size() != 3 || ns.arr.size() != 3) throw std::runtime_error("scene json: triangle needs 3 vertices and normals");
            Vec3 v[3], n[3];
            for (int k = 0; k < 3; ++k) { v[k] = vec3_of(*vs.arr[k]); n[k] = vec3_of(*ns.arr[k]); }
            s.triangles.push_back(make_triangle(v, n, find_material(s, need(*tp, "material").str)));
        }
    if (auto* a = opt_array(root, "meshes"))
        for (auto& mp : *a) {
            const JValue& m = *mp;
            uint32_t material = find_material(s, need(m, "material").str);
            const std::string& wo = need(m, "winding_order").str;
            if (wo != "Clockwise" && wo != "CounterClockwise") throw std::runtime_error("scene json: winding_order");
            std::string path = model_dir.empty() ? need(m, "file_name").str : model_dir + "/" + need(m, "file_name").str;
            std::string obj = read_file(path);
            s.meshes.push_back(load_mesh(need(m, "name").str, obj, material, wo == "CounterClockwise",
                                         s.mesh_triangles, s.mesh_bvh_nodes));
            s.mesh_names.push_back(need(m, "name").str);
        }
    if (auto* a = opt_array(root, "mesh_instances"))
        for (auto& ip : *a) {
            const JValue& in = *ip;
            const std::string& mesh_name = need(in, "mesh").str;
            uint32_t mesh = 0xFFFFFFFFu;
            for (size_t i = 0; i < s.mesh_names.size(); ++i) if (s.mesh_names[i] == mesh_name) { mesh = (uint32_t)i; break; }
            if (mesh == 0xFFFFFFFFu) throw std::runtime_error("scene json: unknown mesh `" + mesh_name + "`");
            int32_t mo = -1;
            const JPtr* ov = in.find("material_override");
            if (ov && (*ov)->kind == JValue::String) mo = (int32_t)find_material(s, (*ov)->str);
            Vec3 od2 = vec3_of(need(in, "orientation_degrees"));
            s.mesh_instances.push_back(make_instance(mesh, vec3_of(need(in, "position")),
                                                     {radians(od2.x), radians(od2.y), radians(od2.z)},
                                                     vec3_of(need(in, "scale")), mo));
        }
    return s;
}

}  // namespace orc
