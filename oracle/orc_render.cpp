// TEST INFRASTRUCTURE - CPU oracle, not part of the shipped renderer.
//
// Per-frame half of the oracle: a line-by-line CPU restatement of the reference's render path,
// lib_raytracer/src/raytracing/{raytracer.rs, intersect.rs}, scene.rs:120-140, utils/mod.rs:9-24,
// raytracing/color.rs:26-52, canvas.rs:35-46 and the pixel loop of native_app/src/main.rs:46-66.
// Recursion, evaluation order and every comparison are kept as written there, NaN behaviour
// included.  Build with -ffp-contract=off, no -ffast-math.
#include "orc_render.h"

#include <limits>
#include <atomic>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

namespace orc {

static const float NUMERIC_ERROR_COMPENSATION_OFFSET = 1e-4f;   // intersect.rs:9
static const float REFLECTION_DIM_FACTOR = 0.8f;                // raytracer.rs:10

// utils/mod.rs:9-11
static inline Vec3 ray_equation(const Ray& ray, float t) { return ray.origin + ray.direction * t; }

// utils/mod.rs:13-24: the first candidate is always taken, later ones only if strictly closer
static inline void take_hitpoint_if_closer(bool& have, Hitpoint& closest, bool cand_valid, const Hitpoint& cand) {
    if (cand_valid) {
        if (have) {
            if (cand.t < closest.t) closest = cand;
        } else {
            closest = cand;
            have = true;
        }
    }
}

// intersect.rs:367-395
static Hitpoint create_hitpoint(float t, Vec3 hit_position, const Ray& ray, Vec3 surface_normal, Vec3 hit_normal,
                                uint32_t material) {
    float n_dot_rdir = dot(surface_normal, ray.direction);
    bool frontside = n_dot_rdir < 0.0f;
    Vec3 sn = frontside ? surface_normal : -surface_normal;
    Vec3 hn = frontside ? hit_normal : -hit_normal;
    Vec3 offset = sn * NUMERIC_ERROR_COMPENSATION_OFFSET;
    Hitpoint h;
    h.t = t;
    h.position = hit_position + offset;
    h.hit_normal = hn;
    h.position_for_refraction = hit_position - offset;
    h.on_frontside = frontside;
    h.material = material;
    h.prim_id = RT_PRIM_NONE;
    h.instance_id = RT_INSTANCE_NONE;
    return h;
}

// intersect.rs:38-112
bool intersect_sphere(const RtSphere& s, const Ray& ray, Hitpoint* out) {
    Vec3 center = v3(s.center);
    float radius = s.radius;
    Vec3 m = ray.origin - center;
    float a = dot(ray.direction, ray.direction);
    float b = 2.0f * dot(m, ray.direction);
    float c = dot(m, m) - radius * radius;

    Vec3 r_dir_norm = normalize(ray.direction);
    Vec3 proj1 = m - r_dir_norm * dot(m, r_dir_norm);
    Vec3 proj2 = m - r_dir_norm * dot(m, r_dir_norm);
    float discriminant = 4.0f * dot(ray.direction, ray.direction) * (radius * radius - dot(proj1, proj2));

    bool have_t = false;
    float t = 0.0f;
    if (discriminant == 0.0f) {                         // is_zero
        t = (-0.5f * b) / a;
        have_t = true;
    } else if (!std::signbit(discriminant)) {           // num_traits Signed::is_positive for f32
        float q = (b < 0.0f) ? -0.5f * (b - std::sqrt(discriminant)) : -0.5f * (b + std::sqrt(discriminant));
        float t0 = q / a;
        float t1 = c / q;
        if (t0 < 0.0f && t1 >= 0.0f) t = t1;
        else if (t1 < 0.0f && t0 >= 0.0f) t = t0;
        else t = fmin32(t0, t1);
        have_t = true;
    }
    if (have_t && t >= 0.0f) {
        Vec3 hit_position = ray_equation(ray, t);
        Vec3 normal = normalize(hit_position - center);   // sphere.rs:12-15
        *out = create_hitpoint(t, hit_position, ray, normal, normal, s.material);
        return true;
    }
    return false;
}

// intersect.rs:114-139
bool intersect_plane(const RtPlane& p, const Ray& ray, Hitpoint* out) {
    Vec3 normal = v3(p.normal);
    float n_dot_rdir = dot(normal, ray.direction);
    bool parallel = n_dot_rdir == 0.0f;
    if (!parallel) {
        float t = (p.distance - dot(normal, ray.origin)) / n_dot_rdir;
        if (t >= 0.0f) {
            Vec3 hit_position = ray_equation(ray, t);
            *out = create_hitpoint(t, hit_position, ray, normal, normal, p.material);
            return true;
        }
    }
    return false;
}

// intersect.rs:141-175
bool intersect_triangle(const RtTriangle& tri, const Ray& ray, Hitpoint* out) {
    Vec3 p0 = v3(tri.vertices[0]), p1 = v3(tri.vertices[1]), p2 = v3(tri.vertices[2]);
    Vec3 e1 = p1 - p0;
    Vec3 e2 = p2 - p0;
    Vec3 q = cross(ray.direction, e2);
    float a = dot(e1, q);
    const float EPSILON = 1e-5f;
    if (a > -EPSILON && a < EPSILON) return false;
    float f = 1.0f / a;
    Vec3 s = ray.origin - p0;
    float u = f * dot(s, q);
    if (u < 0.0f) return false;
    Vec3 r = cross(s, e1);
    float v = f * dot(ray.direction, r);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = f * dot(e2, r);
    if (t < 0.0f) return false;
    float w = 1.0f - u - v;
    Vec3 hit_position = ray_equation(ray, t);
    Vec3 g = w * v3(tri.normals[0]) + u * v3(tri.normals[1]) + v * v3(tri.normals[2]);
    g = normalize(g);
    *out = create_hitpoint(t, hit_position, ray, v3(tri.face_normal), g, tri.material);
    return true;
}

// intersect.rs:185-226
bool intersect_aabb(const float lb[3], const float rt[3], const Ray& ray) {
    float dfx = 1.0f / ray.direction.x;
    float dfy = 1.0f / ray.direction.y;
    float dfz = 1.0f / ray.direction.z;
    float t1 = (lb[0] - ray.origin.x) * dfx;
    float t2 = (rt[0] - ray.origin.x) * dfx;
    float t3 = (lb[1] - ray.origin.y) * dfy;
    float t4 = (rt[1] - ray.origin.y) * dfy;
    float t5 = (lb[2] - ray.origin.z) * dfz;
    float t6 = (rt[2] - ray.origin.z) * dfz;
    float tmin = fmax32(fmax32(fmin32(t1, t2), fmin32(t3, t4)), fmin32(t5, t6));
    float tmax = fmin32(fmin32(fmax32(t1, t2), fmax32(t3, t4)), fmax32(t5, t6));
    if (tmax < 0.0f) return false;
    if (tmin > tmax) return false;
    return true;
}

// intersect.rs:228-269 (and 271-288: USE_BVH is true)
static bool intersect_mesh(const Scene& sc, const RtMesh& mesh, const Ray& ray, Hitpoint* out, Counters* ctr) {
    if (mesh.node_begin == mesh.node_end) return false;
    bool have = false;
    Hitpoint closest{};
    uint32_t stack[RT_BVH_STACK_LEN];
    int sp = 0;
    stack[sp++] = mesh.node_begin;
    while (sp > 0) {
        uint32_t node_index = stack[--sp];
        const RtBvhNode& node = sc.mesh_bvh_nodes[node_index];
        if (ctr) ctr->c[RT_CTR_AABB_TESTS]++;
        if (!intersect_aabb(node.aabb_min, node.aabb_max, ray)) continue;
        if (!node.is_leaf) {
            if (sp + 2 > RT_BVH_STACK_LEN) throw std::runtime_error("BVH traversal stack overflow (ArrayVec capacity 32)");
            stack[sp++] = node.child_left;
            stack[sp++] = node.child_right;
        } else {
            for (uint32_t k = 0; k < node.triangle_count; ++k) {
                uint32_t index = node.triangle_indices[k];
                Hitpoint h;
                if (ctr) ctr->c[RT_CTR_MESH_TRIANGLE_TESTS]++;
                bool ok = intersect_triangle(sc.mesh_triangles[index], ray, &h);
                if (ok) h.prim_id = ((uint32_t)RT_PRIM_MESH_TRIANGLE << 30) | index;
                take_hitpoint_if_closer(have, closest, ok, h);
            }
        }
    }
    if (have) *out = closest;
    return have;
}

static inline Vec3 transform_point(Vec3 v, const float* mat) {
    Mat4 m;
    std::memcpy(m.m, mat, 64);
    Vec4 r = mul(m, Vec4{v.x, v.y, v.z, 1.0f});
    return {r.x, r.y, r.z};
}

// intersect.rs:290-326
static bool intersect_instance(const Scene& sc, uint32_t instance_index, const Ray& ray, Hitpoint* out, Counters* ctr) {
    const RtMeshInstance& inst = sc.mesh_instances[instance_index];
    const RtMesh& mesh = sc.meshes[inst.mesh];
    if (ctr) ctr->c[RT_CTR_INSTANCE_TRANSFORMS]++;
    Ray local;
    local.origin = transform_point(ray.origin, inst.model_inverse);
    local.direction = normalize(transform_point(ray.direction, inst.rotation_scale_inverse));
    Hitpoint h;
    if (!intersect_mesh(sc, mesh, local, &h, ctr)) return false;
    if (ctr) ctr->c[RT_CTR_MESH_HITS]++;
    h.position = transform_point(h.position, inst.model);
    h.hit_normal = normalize(transform_point(h.hit_normal, inst.rotation_scale));
    h.position_for_refraction = transform_point(h.position_for_refraction, inst.model);
    h.t = distance(ray.origin, h.position);
    if (inst.material_override >= 0) h.material = (uint32_t)inst.material_override;
    h.instance_id = instance_index;
    *out = h;
    return true;
}

// scene.rs:120-140 with the slice impl intersect.rs:20-36 for each primitive kind
bool intersect_scene(const Scene& sc, const Ray& ray, Hitpoint* out, Counters* ctr) {
    bool have = false;
    Hitpoint closest{};
    {
        bool g_have = false; Hitpoint g{};
        for (size_t i = 0; i < sc.planes.size(); ++i) {
            Hitpoint h;
            if (ctr) ctr->c[RT_CTR_PLANE_TESTS]++;
            bool ok = intersect_plane(sc.planes[i], ray, &h);
            if (ok) h.prim_id = ((uint32_t)RT_PRIM_PLANE << 30) | (uint32_t)i;
            take_hitpoint_if_closer(g_have, g, ok, h);
        }
        take_hitpoint_if_closer(have, closest, g_have, g);
    }
    {
        bool g_have = false; Hitpoint g{};
        for (size_t i = 0; i < sc.spheres.size(); ++i) {
            Hitpoint h;
            if (ctr) ctr->c[RT_CTR_SPHERE_TESTS]++;
            bool ok = intersect_sphere(sc.spheres[i], ray, &h);
            if (ok) h.prim_id = ((uint32_t)RT_PRIM_SPHERE << 30) | (uint32_t)i;
            take_hitpoint_if_closer(g_have, g, ok, h);
        }
        take_hitpoint_if_closer(have, closest, g_have, g);
    }
    {
        bool g_have = false; Hitpoint g{};
        for (size_t i = 0; i < sc.triangles.size(); ++i) {
            Hitpoint h;
            if (ctr) ctr->c[RT_CTR_TRIANGLE_TESTS]++;
            bool ok = intersect_triangle(sc.triangles[i], ray, &h);
            if (ok) h.prim_id = ((uint32_t)RT_PRIM_TRIANGLE << 30) | (uint32_t)i;
            take_hitpoint_if_closer(g_have, g, ok, h);
        }
        take_hitpoint_if_closer(have, closest, g_have, g);
    }
    for (size_t i = 0; i < sc.mesh_instances.size(); ++i) {
        Hitpoint h;
        bool ok = intersect_instance(sc, (uint32_t)i, ray, &h, ctr);
        take_hitpoint_if_closer(have, closest, ok, h);
    }
    if (have) *out = closest;
    return have;
}

namespace {

struct Tracer {
    const Scene& scene;
    uint32_t max_depth;
    Counters* ctr;
    uint8_t primary_flags = 0;

    bool count_intersect(const Ray& ray, Hitpoint* h, int kind) {
        if (ctr) { ctr->c[RT_CTR_RAYS]++; ctr->c[kind]++; }
        return intersect_scene(scene, ray, h, ctr);
    }

    // raytracer.rs:53-63
    Vec3 trace_background(const Ray& ray) const {
        if (scene.background.type == RT_BACKGROUND_SOLID_COLOR) return v3(scene.background.color);
        Vec3 d = ray.direction + Vec3{1.0f, 1.0f, 1.0f};
        return d * 0.5f;
    }

    // raytracer.rs:190-195
    static Vec3 create_reflected_ray(Vec3 V, Vec3 N) { return 2.0f * dot(N, V) * N - V; }

    // raytracer.rs:197-207
    static Vec3 create_transmitted_ray(Vec3 L, Vec3 N, float n1, float n2) {
        float n = n1 / n2;
        float w = n * dot(L, N);
        float k = std::sqrt(1.0f + (w - n) * (w + n));
        return (w - k) * N - n * L;
    }

    static Vec3 light_world_pos(const RtLight& l) {
        return {l.position[0] / l.position[3], l.position[1] / l.position[3], l.position[2] / l.position[3]};
    }

    // raytracer.rs:209-220
    static Vec3 hitpoint_to_light_unit_vector(const Hitpoint& h, const RtLight& l) {
        bool directional = l.position[3] == 0.0f;
        Vec3 v = directional ? Vec3{l.position[0], l.position[1], l.position[2]} : light_world_pos(l) - h.position;
        return normalize(v);
    }

    // raytracer.rs:222-240
    static float fresnel_reflection(const Ray& reflected, const Ray& transmitted, Vec3 reflection_normal, float n1, float n2) {
        Vec3 transmission_normal = -reflection_normal;
        float cos_i = dot(reflected.direction, reflection_normal);
        float cos_t = dot(transmitted.direction, transmission_normal);
        float n_i = n1, n_t = n2;
        float r_par = (n_t * cos_i - n_i * cos_t) / (n_t * cos_i + n_i * cos_t);
        float r_ort = (n_i * cos_i - n_t * cos_t) / (n_i * cos_i + n_t * cos_t);
        return 0.5f * (r_par * r_par + r_ort * r_ort);
    }

    // raytracer.rs:155-188
    bool trace_shadow_ray(Vec3 world_pos, const RtLight& l) {
        bool directional = l.position[3] == 0.0f;
        Vec3 direction = directional ? Vec3{l.position[0], l.position[1], l.position[2]} : light_world_pos(l) - world_pos;
        direction = normalize(direction);
        Ray ray{world_pos, direction};
        Hitpoint h;
        if (count_intersect(ray, &h, RT_CTR_SHADOW)) {
            if (directional) return true;
            float distance_to_light = distance(ray.origin, light_world_pos(l));
            return h.t < distance_to_light;
        }
        return false;
    }

    // raytracer.rs:137-153
    Vec3 radiance(const Ray& ray, const Hitpoint& h, const RtLight& light, bool is_shadow) {
        if (ctr) ctr->c[RT_CTR_LIGHT_SHADES]++;
        const RtMaterial& m = scene.materials[h.material];
        Vec3 l = hitpoint_to_light_unit_vector(h, light);
        Vec3 n = h.hit_normal;
        Vec3 v = -ray.direction;
        Vec3 r = create_reflected_ray(l, n);
        float l_dot_n = fmax32(dot(l, n), 0.0f);
        float r_dot_v = fmax32(dot(r, v), 0.0f);
        Vec3 emissive = v3(m.emissive);
        Vec3 ambient = cmul(v3(light.ambient), v3(m.ambient));
        Vec3 zero{0.0f, 0.0f, 0.0f};
        Vec3 diffuse = is_shadow ? zero : cmul(v3(light.diffuse), v3(m.diffuse)) * l_dot_n;
        Vec3 specular = is_shadow ? zero : cmul(v3(light.specular), v3(m.specular)) * std::pow(r_dot_v, m.shininess);
        return emissive + ambient + diffuse + specular;
    }

    // raytracer.rs:75-84; returns false for None (no lights)
    bool shade_phong(const Ray& ray, const Hitpoint& h, uint32_t depth, Vec3* out) {
        bool have = false;
        Vec3 color{0, 0, 0};
        for (size_t i = 0; i < scene.lights.size(); ++i) {
            bool is_shadow = trace_shadow_ray(h.position, scene.lights[i]);
            if (depth == 0 && is_shadow && i < 4) primary_flags |= (uint8_t)(RT_FLAG_SHADOW_LIGHT0 << i);
            Vec3 rad = radiance(ray, h, scene.lights[i], is_shadow);
            color = have ? color + rad : rad;     // color::add_option, color.rs:26-36
            have = true;
        }
        *out = color;
        return have;
    }

    // one child of raytracer.rs:90-92 / 109-112: raytrace_impl(..).unwrap_or(trace_background(..))
    Vec3 trace_child(const Ray& child, uint32_t depth, int kind, uint8_t flag_if_primary) {
        Vec3 background = trace_background(child);    // unwrap_or evaluates its argument eagerly
        Vec3 c;
        if (raytrace_impl(child, depth + 1, kind, &c)) {
            if (depth == 0) primary_flags |= flag_if_primary;
            return c;
        }
        return background;
    }

    // raytracer.rs:74-135; returns false for None
    bool shade(const Ray& ray, const Hitpoint& h, uint32_t depth, Vec3* out) {
        const RtMaterial& m = scene.materials[h.material];
        Vec3 V = -ray.direction;
        switch (m.material_type) {
            case RT_MATERIAL_PHONG:
                return shade_phong(ray, h, depth, out);
            case RT_MATERIAL_REFLECT_AND_PHONG: {
                Vec3 direction = create_reflected_ray(V, h.hit_normal);
                Ray reflected{h.position, normalize(direction)};
                Vec3 refl = trace_child(reflected, depth, RT_CTR_REFLECTED, RT_FLAG_REFLECTION_HIT) * REFLECTION_DIM_FACTOR;
                Vec3 phong;
                if (shade_phong(ray, h, depth, &phong)) *out = refl + phong;
                else *out = refl;
                return true;
            }
            default: {   // RT_MATERIAL_REFLECT_AND_REFRACT, raytracer.rs:96-123
                float n1 = h.on_frontside ? m.index_outer : m.index_inner;
                float n2 = h.on_frontside ? m.index_inner : m.index_outer;
                Vec3 dt = create_transmitted_ray(V, h.hit_normal, n1, n2);
                Ray transmitted{h.position_for_refraction, normalize(dt)};
                Vec3 dr = create_reflected_ray(V, h.hit_normal);
                Ray reflected{h.position, normalize(dr)};
                Vec3 reflected_color = trace_child(reflected, depth, RT_CTR_REFLECTED, RT_FLAG_REFLECTION_HIT);
                Vec3 transmitted_color = trace_child(transmitted, depth, RT_CTR_TRANSMITTED, RT_FLAG_REFRACTION_HIT);
                float k_reflected = fresnel_reflection(reflected, transmitted, h.hit_normal, n1, n2);
                float k_transmitted = 1.0f - k_reflected;
                *out = reflected_color * k_reflected + transmitted_color * k_transmitted;
                return true;
            }
        }
    }

    // raytracer.rs:65-72
    bool raytrace_impl(const Ray& ray, uint32_t depth, int kind, Vec3* out, Hitpoint* primary_hit = nullptr) {
        if (depth < max_depth) {
            Hitpoint h;
            if (!count_intersect(ray, &h, kind)) return false;
            if (primary_hit) *primary_hit = h;
            return shade(ray, h, depth, out);
        }
        return false;
    }
};

}  // namespace

// raytracer.rs:30-51
Ray generate_primary_ray(const Scene& s, float x, float y) {
    Vec4 p_screen{x, y, 0.0f, 1.0f};
    Vec4 p_forward{p_screen.x + 0.0f, p_screen.y + 0.0f, p_screen.z + 1.0f, p_screen.w + 0.0f};
    Vec4 pw = mul(s.screen_to_world, p_screen);
    Vec4 pf = mul(s.screen_to_world, p_forward);
    Vec3 a{pw.x / pw.w, pw.y / pw.w, pw.z / pw.w};
    Vec3 b{pf.x / pf.w, pf.y / pf.w, pf.z / pf.w};
    Vec3 direction = b - a;
    return Ray{a, normalize(direction)};
}

// color.rs:42-52: glm::clamp (NaN -> lower bound), *255, `as u8` (truncating, saturating, NaN -> 0)
void quantize(Vec3 c, uint8_t out[3]) {
    float ch[3] = {c.x, c.y, c.z};
    for (int i = 0; i < 3; ++i) {
        float v = ch[i];
        float clamped = (v > 0.0f) ? ((v < 1.0f) ? v : 1.0f) : 0.0f;
        float scaled = clamped * 255.0f;
        int q = (int)scaled;
        if (!(scaled == scaled)) q = 0;
        if (q < 0) q = 0;
        if (q > 255) q = 255;
        out[i] = (uint8_t)q;
    }
}

// native_app/src/main.rs:56-61
PixelResult trace_pixel(const Scene& s, float x, float y, uint32_t max_depth, Counters* ctr) {
    Tracer tr{s, max_depth, ctr};
    Ray ray = generate_primary_ray(s, x, y);
    PixelResult r;
    r.prim_id = RT_PRIM_NONE;
    r.instance_id = RT_INSTANCE_NONE;
    Hitpoint primary{};
    primary.prim_id = RT_PRIM_NONE;
    primary.instance_id = RT_INSTANCE_NONE;
    primary.t = -1.0f;
    Vec3 color;
    // remember whether the primary intersect succeeded even if shade() returns None
    bool hit_recorded = false;
    {
        Hitpoint ph;
        ph.prim_id = RT_PRIM_NONE;
        ph.instance_id = RT_INSTANCE_NONE;
        ph.on_frontside = false;
        bool some = tr.raytrace_impl(ray, 0, RT_CTR_PRIMARY, &color, &ph);
        hit_recorded = ph.prim_id != RT_PRIM_NONE;
        primary = ph;
        if (!some) color = tr.trace_background(ray);
    }
    r.color = color;
    uint8_t flags = tr.primary_flags;
    if (hit_recorded) {
        flags |= RT_FLAG_HIT;
        if (primary.on_frontside) flags |= RT_FLAG_FRONTSIDE;
        r.prim_id = primary.prim_id;
        r.instance_id = primary.instance_id;
    }
    r.flags = flags;
    r.t = hit_recorded ? primary.t : std::numeric_limits<float>::quiet_NaN();
    return r;
}

// Raytracer::depth_map, raytracer.rs:7-8, 21-28: brightness 2^-t * 1.5 of the primary hit; a miss is None, for
// which the pixel loop substitutes the background exactly as it does for raytrace (main.rs:58-61).
Vec3 depth_map_pixel(const Scene& s, float x, float y, Counters* ctr) {
    Tracer tr{s, 1, ctr};
    Ray ray = generate_primary_ray(s, x, y);
    Hitpoint h;
    if (!tr.count_intersect(ray, &h, RT_CTR_PRIMARY)) return tr.trace_background(ray);
    float brightness = std::pow(2.0f, -h.t) * 1.5f;
    return {brightness, brightness, brightness};
}

void render_depth_map(const Scene& s, int threads, int channels, uint8_t* rgb, uint64_t counters[RT_N_COUNTERS]) {
    uint32_t W = s.camera.width, H = s.camera.height;
    if (threads < 1) threads = 1;
    std::atomic<uint32_t> next_row{0};
    std::vector<Counters> per_thread(threads);
    auto worker = [&](int tid) {
        for (;;) {
            uint32_t row = next_row.fetch_add(1);
            if (row >= H) break;
            for (uint32_t x = 0; x < W; ++x) {
                size_t idx = (size_t)row * W + x;
                quantize(depth_map_pixel(s, (float)x, (float)(H - 1 - row), &per_thread[tid]), rgb + idx * channels);
                if (channels == 4) rgb[idx * 4 + 3] = 255;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool) th.join();
    if (counters)
        for (int k = 0; k < RT_N_COUNTERS; ++k) {
            uint64_t sum = 0;
            for (auto& c : per_thread) sum += c.c[k];
            counters[k] = sum;
        }
}

// native_app/src/main.rs:46-66 + canvas.rs:35-46 (canvas row r shows screen y = H-1-r)
void render_frame(const Scene& s, uint32_t max_depth, int threads, uint32_t row_begin, uint32_t row_end, int channels,
                  uint8_t* rgb, uint32_t* prim_id, uint32_t* instance_id, uint8_t* flags, float* hit_t,
                  uint64_t counters[RT_N_COUNTERS]) {
    uint32_t W = s.camera.width, H = s.camera.height;
    if (row_end > H) row_end = H;
    if (threads < 1) threads = 1;
    std::atomic<uint32_t> next_row{row_begin};
    std::vector<Counters> per_thread(threads);
    auto worker = [&](int tid) {
        Counters& ctr = per_thread[tid];
        for (;;) {
            uint32_t row = next_row.fetch_add(1);
            if (row >= row_end) break;
            uint32_t y = H - 1 - row;
            for (uint32_t x = 0; x < W; ++x) {
                PixelResult pr = trace_pixel(s, (float)x, (float)y, max_depth, &ctr);
                size_t idx = (size_t)row * W + x;
                if (rgb) {
                    quantize(pr.color, rgb + idx * channels);
                    if (channels == 4) rgb[idx * 4 + 3] = 255;
                }
                if (prim_id) prim_id[idx] = pr.prim_id;
                if (instance_id) instance_id[idx] = pr.instance_id;
                if (flags) flags[idx] = pr.flags;
                if (hit_t) hit_t[idx] = pr.t;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool) th.join();
    if (counters) {
        for (int k = 0; k < RT_N_COUNTERS; ++k) {
            uint64_t sum = 0;
            for (auto& c : per_thread) sum += c.c[k];
            counters[k] = sum;
        }
    }
}

}  // namespace orc
