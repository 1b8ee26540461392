// TEST INFRASTRUCTURE - CPU oracle, not part of the shipped renderer.
//
// C entry points of the oracle for ctypes (tests/, __graft_entry__.smoke() and the cpu_baseline
// leg of bench.py are the only callers).
#include <cstring>
#include <exception>
#include <string>

#include "orc_render.h"

using namespace orc;

static thread_local std::string g_error;

extern "C" {

const char* orc_last_error() { return g_error.c_str(); }

void* orc_scene_load_text(const char* json_text, const char* model_dir) {
    try {
        return new Scene(parse_scene_json(json_text, model_dir ? model_dir : ""));
    } catch (const std::exception& e) {
        g_error = e.what();
        return nullptr;
    }
}

void* orc_scene_load_file(const char* json_path, const char* model_dir) {
    try {
        return new Scene(parse_scene_json(read_file(json_path), model_dir ? model_dir : ""));
    } catch (const std::exception& e) {
        g_error = e.what();
        return nullptr;
    }
}

// Builds an oracle scene from flat arrays (copied).  The camera matrix is taken as given.
void* orc_scene_from_desc(const RtSceneDesc* d, const float screen_to_world[16], uint32_t width, uint32_t height) {
    try {
        auto* s = new Scene();
        s->background = d->background;
        s->lights.assign(d->lights, d->lights + d->n_lights);
        s->materials.assign(d->materials, d->materials + d->n_materials);
        s->planes.assign(d->planes, d->planes + d->n_planes);
        s->spheres.assign(d->spheres, d->spheres + d->n_spheres);
        s->triangles.assign(d->triangles, d->triangles + d->n_triangles);
        s->mesh_triangles.assign(d->mesh_triangles, d->mesh_triangles + d->n_mesh_triangles);
        s->mesh_bvh_nodes.assign(d->mesh_bvh_nodes, d->mesh_bvh_nodes + d->n_mesh_bvh_nodes);
        s->meshes.assign(d->meshes, d->meshes + d->n_meshes);
        s->mesh_instances.assign(d->mesh_instances, d->mesh_instances + d->n_mesh_instances);
        std::memcpy(s->screen_to_world.m, screen_to_world, 64);
        s->camera.width = width;
        s->camera.height = height;
        return s;
    } catch (const std::exception& e) {
        g_error = e.what();
        return nullptr;
    }
}

void orc_scene_free(void* h) { delete static_cast<Scene*>(h); }

void orc_scene_resize(void* h, uint32_t width, uint32_t height) { resize_screen(*static_cast<Scene*>(h), width, height); }

void orc_scene_turn_camera(void* h, float bx, float by, float ex, float ey) {
    turn_camera(*static_cast<Scene*>(h), bx, by, ex, ey);
}

void orc_scene_set_camera_matrix(void* h, const float m[16], uint32_t width, uint32_t height) {
    Scene& s = *static_cast<Scene*>(h);
    std::memcpy(s.screen_to_world.m, m, 64);
    s.camera.width = width;
    s.camera.height = height;
}

void orc_scene_camera(void* h, float screen_to_world[16], uint32_t dims[2]) {
    const Scene& s = *static_cast<Scene*>(h);
    std::memcpy(screen_to_world, s.screen_to_world.m, 64);
    dims[0] = s.camera.width;
    dims[1] = s.camera.height;
}

// Pointers into the oracle scene; valid until the scene is freed or reloaded.
void orc_scene_desc(void* h, RtSceneDesc* d) {
    const Scene& s = *static_cast<Scene*>(h);
    std::memset(d, 0, sizeof(*d));
    d->background = s.background;
    d->n_lights = (uint32_t)s.lights.size();                 d->lights = s.lights.data();
    d->n_materials = (uint32_t)s.materials.size();           d->materials = s.materials.data();
    d->n_planes = (uint32_t)s.planes.size();                 d->planes = s.planes.data();
    d->n_spheres = (uint32_t)s.spheres.size();               d->spheres = s.spheres.data();
    d->n_triangles = (uint32_t)s.triangles.size();           d->triangles = s.triangles.data();
    d->n_mesh_triangles = (uint32_t)s.mesh_triangles.size(); d->mesh_triangles = s.mesh_triangles.data();
    d->n_mesh_bvh_nodes = (uint32_t)s.mesh_bvh_nodes.size(); d->mesh_bvh_nodes = s.mesh_bvh_nodes.data();
    d->n_meshes = (uint32_t)s.meshes.size();                 d->meshes = s.meshes.data();
    d->n_mesh_instances = (uint32_t)s.mesh_instances.size(); d->mesh_instances = s.mesh_instances.data();
}

int orc_render(void* h, uint32_t max_depth, int threads, uint32_t row_begin, uint32_t row_end, int channels,
               uint8_t* rgb, uint32_t* prim_id, uint32_t* instance_id, uint8_t* flags, float* hit_t, uint64_t* counters) {
    try {
        render_frame(*static_cast<Scene*>(h), max_depth, threads, row_begin, row_end, channels, rgb, prim_id,
                     instance_id, flags, hit_t, counters);
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

int orc_render_depth_map(void* h, int threads, int channels, uint8_t* rgb, uint64_t* counters) {
    try {
        render_depth_map(*static_cast<Scene*>(h), threads, channels, rgb, counters);
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

// Unquantised colour of one pixel; (x, y) in the reference's screen coordinates (y up).
int orc_trace_pixel(void* h, float x, float y, uint32_t max_depth, float color[3], uint32_t ids[2], uint8_t* flags) {
    try {
        PixelResult r = trace_pixel(*static_cast<Scene*>(h), x, y, max_depth, nullptr);
        color[0] = r.color.x; color[1] = r.color.y; color[2] = r.color.z;
        ids[0] = r.prim_id; ids[1] = r.instance_id;
        *flags = r.flags;
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

// ---- single-primitive entry points for the reference's known-answer tests -------------------

static Ray mk_ray(const float o[3], const float d[3]) { return Ray{v3(o), v3(d)}; }

int orc_kat_sphere(const float center[3], float radius, const float o[3], const float d[3], float* t) {
    RtSphere s{};
    std::memcpy(s.center, center, 12);
    s.radius = radius;
    Hitpoint h;
    if (!intersect_sphere(s, mk_ray(o, d), &h)) return 0;
    *t = h.t;
    return 1;
}

int orc_kat_plane(const float normal[3], float distance, const float o[3], const float d[3], float* t) {
    RtPlane p = make_plane(v3(normal), distance, 0);
    Hitpoint h;
    if (!intersect_plane(p, mk_ray(o, d), &h)) return 0;
    *t = h.t;
    return 1;
}

int orc_kat_triangle(const float vertices[9], const float normals[9], const float o[3], const float d[3], float* t) {
    Vec3 v[3] = {v3(vertices), v3(vertices + 3), v3(vertices + 6)};
    Vec3 n[3] = {v3(normals), v3(normals + 3), v3(normals + 6)};
    RtTriangle tri = make_triangle(v, n, 0);
    Hitpoint h;
    if (!intersect_triangle(tri, mk_ray(o, d), &h)) return 0;
    *t = h.t;
    return 1;
}

int orc_kat_aabb(const float bmin[3], const float bmax[3], const float o[3], const float d[3]) {
    return intersect_aabb(bmin, bmax, mk_ray(o, d)) ? 1 : 0;
}

int orc_hull(const float* vertices, uint32_t n, float out_min_max[6]) {
    Aabb b;
    if (!hull_from_vertices(reinterpret_cast<const Vec3*>(vertices), n, &b)) return 0;
    out_min_max[0] = b.min.x; out_min_max[1] = b.min.y; out_min_max[2] = b.min.z;
    out_min_max[3] = b.max.x; out_min_max[4] = b.max.y; out_min_max[5] = b.max.z;
    return 1;
}

float orc_aabb_surface_area(const float bmin[3], const float bmax[3]) { return aabb_surface_area(Aabb{v3(bmin), v3(bmax)}); }

void orc_mat_inverse(const float in[16], float out[16]) {
    Mat4 m;
    std::memcpy(m.m, in, 64);
    Mat4 r = inverse(m);
    std::memcpy(out, r.m, 64);
}

void orc_mat_mul(const float a[16], const float b[16], float out[16]) {
    Mat4 x, y;
    std::memcpy(x.m, a, 64);
    std::memcpy(y.m, b, 64);
    Mat4 r = mul(x, y);
    std::memcpy(out, r.m, 64);
}

void orc_make_instance(uint32_t mesh, const float position[3], const float orientation_radians[3], const float scale[3],
                       int32_t material_override, RtMeshInstance* out) {
    *out = make_instance(mesh, v3(position), v3(orientation_radians), v3(scale), material_override);
}

// camera = position[3], orientation_radians[3], y_fov_degrees, z_near, z_far
void orc_screen_to_world(const float camera[9], uint32_t width, uint32_t height, float out[16]) {
    Camera c;
    c.position = v3(camera);
    c.orientation = v3(camera + 3);
    c.y_fov_degrees = camera[6];
    c.z_near = camera[7];
    c.z_far = camera[8];
    c.width = width;
    c.height = height;
    Mat4 m = m_screen_to_world(c);
    std::memcpy(out, m.m, 64);
}

// Loads an OBJ text as one mesh and returns the triangle / node counts; arrays are written if
// non-null and large enough (call twice).  Used to cross-check the shipped loader's BVH.
int orc_build_mesh(const char* obj_text, int counter_clockwise, uint32_t material, RtTriangle* tris, uint32_t tri_cap,
                   RtBvhNode* nodes, uint32_t node_cap, RtMesh* mesh) {
    try {
        std::vector<RtTriangle> t;
        std::vector<RtBvhNode> n;
        *mesh = load_mesh("mesh", obj_text, material, counter_clockwise != 0, t, n);
        if (tris && tri_cap >= t.size()) std::memcpy(tris, t.data(), t.size() * sizeof(RtTriangle));
        if (nodes && node_cap >= n.size()) std::memcpy(nodes, n.data(), n.size() * sizeof(RtBvhNode));
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

}  // extern "C"
