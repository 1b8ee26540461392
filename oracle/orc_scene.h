// TEST INFRASTRUCTURE - CPU oracle, not part of the shipped renderer.
//
// Scene container of the oracle.  The arrays are kept in the layout of include/rt_hip.h so that
// a test can hand the very same bytes to the HIP renderer (orc_scene_desc in orc_api.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../include/rt_hip.h"
#include "orc_math.h"

namespace orc {

// raytracing/camera.rs:3-11 (orientation already in radians, parser.rs:40)
struct Camera {
    Vec3 position{0, 0, 0};
    Vec3 orientation{0, 0, 0};
    uint32_t width = 0, height = 0;
    float y_fov_degrees = 0, z_near = 0, z_far = 0;
};

// scene.rs:6-20
struct Scene {
    Camera camera;
    Mat4 screen_to_world{};
    RtBackground background{};
    std::vector<RtLight> lights;
    std::vector<RtMaterial> materials;
    std::vector<std::string> material_names;
    std::vector<RtPlane> planes;
    std::vector<RtSphere> spheres;
    std::vector<RtTriangle> triangles;
    std::vector<RtTriangle> mesh_triangles;
    std::vector<RtBvhNode> mesh_bvh_nodes;
    std::vector<RtMesh> meshes;
    std::vector<std::string> mesh_names;
    std::vector<RtMeshInstance> mesh_instances;
};

// transform.rs
Mat4 m_translation(Vec3 offset);
Mat4 m_scaling(Vec3 scale);
Mat4 m_rotation(float yaw, float pitch, float roll);
Mat4 m_model(Vec3 position, Vec3 orientation, Vec3 scale);
Mat4 m_viewport(float x, float y, float w, float h, float z_near, float z_far);
Mat4 m_projection(float y_fov_degrees, float aspect, float z_near, float z_far);
Mat4 m_view(Vec3 orientation, Vec3 position);
Mat4 m_screen_to_world(const Camera& camera);

// scene.rs:54-58, 60-97
void resize_screen(Scene& scene, uint32_t width, uint32_t height);
void turn_camera(Scene& scene, float begin_x, float begin_y, float end_x, float end_y);

// geometry constructors
RtPlane make_plane(Vec3 normal, float distance, uint32_t material);                        // plane.rs:15-23
RtTriangle make_triangle(const Vec3 vertices[3], const Vec3 normals[3], uint32_t material);  // triangle.rs:15-44
Vec3 triangle_centroid(const RtTriangle& t);                                                 // triangle.rs:29-36
RtMeshInstance make_instance(uint32_t mesh, Vec3 position, Vec3 orientation, Vec3 scale,
                             int32_t material_override);                                    // instance.rs:17-39

// hull.rs / aabb.rs
struct Aabb { Vec3 min, max; };
bool hull_from_vertices(const Vec3* vertices, size_t n, Aabb* out);   // hull.rs:17-55; false = None
float aabb_surface_area(const Aabb& b);                                // aabb.rs:9-19

// object_file.rs:18-61.  `obj_text` is the OBJ file's contents.  Appends to the scene's mesh
// triangle and BVH node arrays and returns the mesh record; throws std::runtime_error.
RtMesh load_mesh(const std::string& name, const std::string& obj_text, uint32_t material,
                 bool counter_clockwise, std::vector<RtTriangle>& mesh_triangles,
                 std::vector<RtBvhNode>& mesh_bvh_nodes);

// bvh/builder.rs:32-68
RtMesh build_bvh(uint32_t tri_begin, uint32_t tri_end, const std::vector<RtTriangle>& mesh_triangles,
                 std::vector<RtBvhNode>& mesh_bvh_nodes);

// scene_file/parser.rs:61-78.  `model_dir` is where mesh file names are looked up
// (native_app/src/filesystem_mesh_loader.rs:16-30).  Throws std::runtime_error.
Scene parse_scene_json(const std::string& json_text, const std::string& model_dir);

std::string read_file(const std::string& path);

}  // namespace orc
