/*
 * rt_host.h - setup-time half of the drop-in: builds the arrays of include/rt_hip.h's RtSceneDesc
 * from the reference's own inputs (a scene JSON file and OBJ meshes) without a Rust toolchain.
 *
 * In the real drop-in flow the Rust crate keeps doing this work (scene_file/parser.rs,
 * object_file.rs, bvh/builder.rs, raytracing/transform.rs) and hands its Scene to rt_scene_create.
 * These entry points exist so that C, C++ and Python callers - the benchmark and the tests of this
 * repository among them - can assemble the same scenes.  Pure CPU code; no GPU is needed.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RtHostScene RtHostScene;

/* The reference's MeshLoader trait (scene_file/mesh_loader.rs:5-11) as a callback: given the
 * "file_name" of a scene's mesh entry, return the OBJ file's bytes in *out_data / *out_len and a
 * release function (may be NULL) that is called with the same pointers once they have been parsed.
 * Return 0 on success, non-zero to fail the load (the reference unwraps the io::Error,
 * parser.rs:202-203). */
typedef int (*RtMeshLoadFn)(void* user, const char* file_name, const char** out_data, size_t* out_len,
                            void (**out_release)(void* user, const char* data, size_t len));

/* Parser::parse_json (scene_file/parser.rs:61-78) with a caller-supplied mesh loader. */
int rt_host_scene_parse(const char* json_text, size_t json_len, RtMeshLoadFn loader, void* loader_user,
                        RtHostScene** out_scene);

/* The same with native_app's FilesystemMeshLoader (native_app/src/filesystem_mesh_loader.rs:11-30):
 * mesh files are read from `model_dir`. */
int rt_host_scene_load(const char* json_path, const char* model_dir, RtHostScene** out_scene);

/* Borrowed view of the scene's arrays; valid until the scene is destroyed. */
int rt_host_scene_desc(const RtHostScene* scene, RtSceneDesc* out_desc);

/* Scene::screen_to_world() (scene.rs:50-52) and camera().screen_dimensions. */
int rt_host_scene_camera(const RtHostScene* scene, float screen_to_world[16], uint32_t* width, uint32_t* height);

/* Scene::resize_screen (scene.rs:54-58) */
int rt_host_scene_resize_screen(RtHostScene* scene, uint32_t width, uint32_t height);

/* Scene::turn_camera (scene.rs:60-97); pixel coordinates of a drag */
int rt_host_scene_turn_camera(RtHostScene* scene, float begin_x, float begin_y, float end_x, float end_y);

/* Scene::update_camera (scene.rs:45-48) for the two fields callers move: position and orientation
 * (radians: x pitch, y yaw, z roll).  Either pointer may be NULL to keep the current value. */
int rt_host_scene_set_camera_pose(RtHostScene* scene, const float position[3], const float orientation[3]);
int rt_host_scene_get_camera_pose(const RtHostScene* scene, float position[3], float orientation[3]);

/* Index of the material / mesh with this name (parser.rs:124-144), or -1. */
int rt_host_scene_find_material(const RtHostScene* scene, const char* name);
int rt_host_scene_find_mesh(const RtHostScene* scene, const char* name);

/* Instance::new (raytracing/instance.rs:17-39) appended to the scene: lets callers build the
 * synthetic many-instance scenes without writing JSON.  Orientation in radians. */
int rt_host_scene_add_instance(RtHostScene* scene, uint32_t mesh, const float position[3], const float orientation[3],
                               const float scale[3], int32_t material_override);

/* object_file::load_mesh (object_file.rs:18-61) appended to the scene under `name`. */
int rt_host_scene_add_mesh(RtHostScene* scene, const char* name, const char* obj_data, size_t obj_len,
                           uint32_t material, int counter_clockwise);

void rt_host_scene_destroy(RtHostScene* scene);

/* Replaces `impl WritePng for Canvas` (native_app/src/write_png.rs:8-30): an 8-bit PNG of a canvas as rt_render
 * fills it (row 0 = top), colour type RGB for channels == 3 (what the reference writes) or RGBA for 4.  The
 * image data is stored uncompressed (valid deflate "stored" blocks): any PNG reader decodes the same pixels. */
int rt_host_write_png(const char* path, const uint8_t* pixels, uint32_t width, uint32_t height, uint32_t channels);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
