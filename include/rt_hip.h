/*
 * rt_hip.h - C ABI of the MI355X (gfx950) frame renderer.
 *
 * The library replaces lib_raytracer's per-frame render path: the pixel loop of
 * native_app/src/main.rs:46-66 together with everything it calls in
 * lib_raytracer/src/raytracing/raytracer.rs, raytracing/intersect.rs and
 * scene.rs:120-140.  The shape of the interface follows the reference's one
 * frame-level renderer, GpuRenderer (web_app/wasm/src/gpu_renderer.rs:44-68,
 * 280, 356-364): the scene is uploaded once, the camera is rewritten per frame,
 * and `render` fills a caller-owned 8-bit canvas whose row 0 is the top row.
 *
 * All scene inputs are plain arrays of the structs below.  They mirror the
 * public fields of lib_raytracer::Scene (scene.rs:6-20) one to one; material
 * and mesh references are the indices the reference's parser has already
 * resolved (scene_file/parser.rs:124-144).  Matrices are column-major, 16
 * floats, exactly as nalgebra stores a glm::Mat4.
 *
 * Conventions: every input pointer is borrowed for the duration of the call;
 * outputs are caller-allocated; functions return RT_OK (0) or a negative
 * RtStatus and never unwind; a handle must be driven by one thread at a time.
 * The text of the last error of the calling thread is available from
 * rt_last_error().
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum RtStatus {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1,
    RT_ERR_OUT_OF_RANGE = -2,   /* an index in the scene points outside its array */
    RT_ERR_BVH_TOO_DEEP = -3,   /* traversal would need more than RT_BVH_STACK_LEN entries */
    RT_ERR_DEVICE = -4,         /* a HIP call failed */
    RT_ERR_NO_DEVICE = -5,
    RT_ERR_IO = -6,
    RT_ERR_PARSE = -7
} RtStatus;

/* raytracing/material.rs:18-25; numbering as gpu_types.rs:196-201 */
enum { RT_MATERIAL_PHONG = 0, RT_MATERIAL_REFLECT_AND_PHONG = 1, RT_MATERIAL_REFLECT_AND_REFRACT = 2 };
/* raytracing/background.rs:3-7; numbering as gpu_types.rs:32-45 */
enum { RT_BACKGROUND_SOLID_COLOR = 0, RT_BACKGROUND_COLORED_DIRECTION = 1 };

/* intersect.rs:240 - the reference's traversal stack holds 32 node indices and panics beyond. */
#define RT_BVH_STACK_LEN 32
/* bvh/node.rs:22 */
#define RT_LEAF_TRIANGLES 5
/* raytracer.rs:9 */
#define RT_DEFAULT_MAX_DEPTH 10

/* raytracing/light.rs:4-13.  position.w == 0 marks a directional light. */
typedef struct RtLight {
    float position[4];
    float ambient[3];
    float diffuse[3];
    float specular[3];
} RtLight;

/* raytracing/material.rs:6-16 */
typedef struct RtMaterial {
    float emissive[3];
    float ambient[3];
    float diffuse[3];
    float specular[3];
    float shininess;
    uint32_t material_type;   /* RT_MATERIAL_* */
    float index_inner;        /* only for RT_MATERIAL_REFLECT_AND_REFRACT */
    float index_outer;
} RtMaterial;

/* geometry/plane.rs:6-12; `normal` is already normalised by Plane::new */
typedef struct RtPlane {
    float normal[3];
    float distance;
    uint32_t material;
} RtPlane;

/* geometry/sphere.rs:4-9 */
typedef struct RtSphere {
    float center[3];
    float radius;
    uint32_t material;
} RtSphere;

/* geometry/triangle.rs:6-12; used both for Scene::triangles and Scene::mesh_triangles */
typedef struct RtTriangle {
    float vertices[3][3];
    float normals[3][3];
    float face_normal[3];     /* Triangle::normal(), precomputed by Triangle::new */
    uint32_t material;
} RtTriangle;

/* bvh/node.rs:6-19 + geometry/aabb.rs:3-6.  Child and triangle indices are absolute
 * indices into RtSceneDesc::mesh_bvh_nodes / mesh_triangles, as in the reference. */
typedef struct RtBvhNode {
    float aabb_min[3];
    float aabb_max[3];
    uint32_t is_leaf;
    uint32_t child_left;
    uint32_t child_right;
    uint32_t triangle_count;                       /* <= RT_LEAF_TRIANGLES */
    uint32_t triangle_indices[RT_LEAF_TRIANGLES];
} RtBvhNode;

/* geometry/mesh.rs:8-12 + bvh/mod.rs:12-15; ranges are half-open [begin, end) */
typedef struct RtMesh {
    uint32_t triangle_begin;
    uint32_t triangle_end;
    uint32_t node_begin;      /* the root is node_begin (intersect.rs:243) */
    uint32_t node_end;
    uint32_t max_depth;
} RtMesh;

/* raytracing/instance.rs:6-14 */
typedef struct RtMeshInstance {
    float rotation_scale[16];
    float rotation_scale_inverse[16];
    float model[16];
    float model_inverse[16];
    uint32_t mesh;
    int32_t material_override;    /* -1 = None */
} RtMeshInstance;

typedef struct RtBackground {
    uint32_t type;                /* RT_BACKGROUND_* */
    float color[3];               /* only for RT_BACKGROUND_SOLID_COLOR */
} RtBackground;

/* The arrays of lib_raytracer::Scene (scene.rs:6-20), borrowed. */
typedef struct RtSceneDesc {
    RtBackground background;
    uint32_t n_lights;
    uint32_t n_materials;
    uint32_t n_planes;
    uint32_t n_spheres;
    uint32_t n_triangles;
    uint32_t n_mesh_triangles;
    uint32_t n_mesh_bvh_nodes;
    uint32_t n_meshes;
    uint32_t n_mesh_instances;
    const RtLight* lights;
    const RtMaterial* materials;
    const RtPlane* planes;
    const RtSphere* spheres;
    const RtTriangle* triangles;
    const RtTriangle* mesh_triangles;
    const RtBvhNode* mesh_bvh_nodes;
    const RtMesh* meshes;
    const RtMeshInstance* mesh_instances;
} RtSceneDesc;

/* Per-pixel primary-hit record and frame-wide ray statistics (optional outputs). */

/* prim_id = kind << 30 | index.  For RT_PRIM_MESH_TRIANGLE the index is the absolute
 * index into mesh_triangles and instance_id names the Scene::mesh_instances entry. */
enum { RT_PRIM_PLANE = 0, RT_PRIM_SPHERE = 1, RT_PRIM_TRIANGLE = 2, RT_PRIM_MESH_TRIANGLE = 3 };
#define RT_PRIM_NONE 0xFFFFFFFFu
#define RT_INSTANCE_NONE 0xFFFFFFFFu

/* bits of RtAux::flags, all about the primary ray of the pixel */
#define RT_FLAG_HIT            0x01u  /* Scene::intersect returned Some */
#define RT_FLAG_FRONTSIDE      0x02u  /* Hitpoint::on_frontside */
#define RT_FLAG_REFLECTION_HIT 0x04u  /* the reflected child ray (raytracer.rs:91,109) hit geometry */
#define RT_FLAG_REFRACTION_HIT 0x08u  /* the transmitted child ray (raytracer.rs:111) hit geometry */
#define RT_FLAG_SHADOW_LIGHT0  0x10u  /* trace_shadow_ray (raytracer.rs:78) said "shadow" for light i: bit 4+i, i < 4 */

enum {
    RT_CTR_RAYS = 0,          /* calls of Scene::intersect = primary + shadow + reflected + transmitted */
    RT_CTR_PRIMARY,
    RT_CTR_SHADOW,
    RT_CTR_REFLECTED,
    RT_CTR_TRANSMITTED,
    RT_CTR_AABB_TESTS,        /* AABB::intersect calls of the reference's un-pruned traversal */
    RT_CTR_MESH_TRIANGLE_TESTS,
    RT_CTR_TRIANGLE_TESTS,    /* Scene::triangles */
    RT_CTR_SPHERE_TESTS,
    RT_CTR_PLANE_TESTS,
    RT_CTR_INSTANCE_TRANSFORMS,
    RT_CTR_MESH_HITS,         /* instance traversals that returned a hit (normals fetched) */
    RT_CTR_LIGHT_SHADES,      /* Raytracer::radiance calls */
    /* what the device's own BVH walk did (pruned, 4-wide tree): nodes it visited - four box tests each - and triangle
     * tests it ran; no counterpart in the reference's walk, which the five counters above describe */
    RT_CTR_WALK_NODE_VISITS,
    RT_CTR_WALK_TRIANGLE_TESTS,
    /* (ray, instance) pairs whose ray enters the mesh so far out that it passes nearly every box: tested against every
     * leaf by a block of their own instead of being walked (their box and triangle tests are in the two counters above) */
    RT_CTR_WALK_HEAVY_PAIRS,
    /* shadow rays (counted in RT_CTR_SHADOW and RT_CTR_RAYS like every ray the reference traces) that were not traced because
     * their verdict could not change a bit of the picture: radiance() (raytracer.rs:137-153) only uses it to switch the
     * diffuse and the specular term off, and both were zero already - the surface faces away from the light and the
     * mirrored light direction away from the viewer, or the colour products are zero */
    RT_CTR_SHADOW_NOT_TRACED,
    RT_N_COUNTERS
};

typedef struct RtAux {              /* with several views (rt_scene_set_views): n_views * W*H entries per array */
    uint32_t* prim_id;        /* W*H, row 0 = top, or NULL */
    uint32_t* instance_id;    /* W*H or NULL */
    uint8_t* flags;           /* W*H or NULL */
    uint64_t counters[RT_N_COUNTERS];   /* the device path fills RT_CTR_RAYS..RT_CTR_TRANSMITTED and RT_CTR_WALK_*, the rest stay 0 */
    float* hit_t;             /* W*H or NULL: Hitpoint::t of the primary hit (hitpoint.rs:5), NaN where there is none */
} RtAux;

enum { RT_FORMAT_RGB8 = 0, RT_FORMAT_RGBA8 = 1 };

/* Frame tiles are RT_TILE_DIM x RT_TILE_DIM pixels, numbered row-major over the tile grid
 * (ceil(W/RT_TILE_DIM) columns).  Shard s of n renders the tiles t with t % n == s - the
 * row interlacing of web_app/wasm/src/renderer.rs:68-76, in two dimensions. */
#define RT_TILE_DIM 32

/* What a pixel shows: Raytracer::raytrace (raytracer.rs:17-19) or Raytracer::depth_map (raytracer.rs:21-28,
 * brightness 2^-t * 1.5 of the primary hit, the background where there is none). */
enum { RT_MODE_RAYTRACE = 0, RT_MODE_DEPTH_MAP = 1 };

typedef struct RtRenderParams {
    uint32_t max_depth;       /* MAX_RAY_RECURSION_DEPTH, raytracer.rs:9; 0 means RT_DEFAULT_MAX_DEPTH */
    uint32_t output_format;   /* RT_FORMAT_* ; RGBA8 sets alpha = 255 as web_app/wasm/src/color.rs:10-21 */
    uint32_t shard_index;     /* which tiles to render: see RT_TILE_DIM */
    uint32_t shard_count;     /* 0 or 1 = the whole frame */
    uint32_t mode;            /* RT_MODE_* */
    uint32_t reserved;        /* must be 0 */
    void* stream;             /* hipStream_t to enqueue on; NULL = the handle's own stream */
} RtRenderParams;

typedef struct RtScene RtScene;

/* Replaces GpuRenderer::new / setup_pipeline_and_buffers (gpu_renderer.rs:70-246): validates the
 * scene, copies it into device-resident arrays on HIP device `device_id`.  The camera is unset. */
int rt_scene_create(const RtSceneDesc* desc, int device_id, RtScene** out_scene);

/* Replaces the per-frame camera-uniform write (gpu_renderer.rs:321-322).  `screen_to_world` is
 * Scene::screen_to_world() (scene.rs:50-52), column-major; width/height are camera.screen_dimensions. */
int rt_scene_set_camera(RtScene* scene, const float screen_to_world[16], uint32_t width, uint32_t height);

/* Several views of the scene rendered by one call, all of the same size: `screen_to_world` holds n_views matrices
 * (16 floats each) back to back, 1 <= n_views <= RT_MAX_VIEWS.  rt_render / rt_render_device then produce n_views
 * frames (or shards) back to back in the output, view v at byte offset v * (size of one frame or shard); the RtAux
 * arrays hold n_views * W*H entries; rt_assemble_shards expects every rank's buffer to hold its n_views shards and
 * writes n_views frames; the ray counts and times are those of all views together.  The views travel through every
 * recursion depth together, so the fixed cost of a depth (launches, the longest BVH walk) is paid once for all of
 * them: small frames and shards render markedly faster per view this way.  rt_scene_set_camera is the case n_views = 1.
 * A handle keeps the buffers of the largest frame it has rendered, so going back and forth between view counts
 * (a last, smaller pass of a batch) allocates nothing.
 * The reference has no counterpart: its renderers draw one camera per call (gpu_renderer.rs:280, main.rs:46-66);
 * a caller with a queue of frames (an animation, the batch of orbit views) would loop over it. */
#define RT_MAX_VIEWS 256     /* and n_views * (pixels of a view, or of a shard of it) <= 2^28 */
int rt_scene_set_views(RtScene* scene, const float* screen_to_world, uint32_t n_views, uint32_t width, uint32_t height);

/* Replaces paint_scene (native_app/src/main.rs:46-66) and GpuRenderer::render (gpu_renderer.rs:280-354):
 * renders the whole frame (shard fields of `params` are ignored) and copies it to `out_pixels`
 * (host memory, W*H*3 or W*H*4 bytes, row 0 = top as canvas.rs:35-46).  `aux` may be NULL; its
 * non-NULL arrays are host memory.  Synchronous. */
int rt_render(RtScene* scene, const RtRenderParams* params, uint8_t* out_pixels, RtAux* aux);

/* Same frame, output left in device memory and the call asynchronous on params->stream.
 * shard_count <= 1: `d_out` receives the row-major frame.  shard_count > 1: `d_out` receives only
 * this shard's tiles, tile-major: local tile k (global tile shard_index + k*shard_count) occupies
 * bytes [k*T, (k+1)*T), T = RT_TILE_DIM*RT_TILE_DIM*bytes_per_pixel, pixels row-major inside the
 * tile; pixels outside the frame are zero.  Use rt_shard_bytes for the buffer size. */
int rt_render_device(RtScene* scene, const RtRenderParams* params, void* d_out);

/* rt_render_device without any allocation.  No counterpart in the reference: it exists for callers that cannot
 * allocate device memory at the time they render - code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  Same arguments, same output bytes as rt_render_device, but the frame's work buffers (ray lists,
 * candidate keys, BVH work items, hit records, colour sums, counters) are carved out of one static array in device
 * memory that the library holds on every GPU it runs on (rt_noalloc_scratch_bytes: 128 MiB) instead of the handle's
 * own buffers, and the call performs no hipMalloc / hipFree, whatever the frame size or the handle's history.
 * Limits: the frame's buffers must fit the array - about 2.3 KB per local pixel (views x this shard's tiles) in a scene
 * with two lights, i.e. frames up to roughly 256x192; larger ones fail with RT_ERR_OUT_OF_RANGE and a message that
 * names both sizes (use rt_render_device for them).  The lists hold eight rays per pixel and recursion depth and
 * cannot grow: if a frame needs more, rt_last_ray_counts reports the overflow as it does after rt_render_device.
 * Kernel timing mode is ignored.  All calls on one GPU - from any handle, on any stream - share the one array:
 * calls that overlap in time race and corrupt each other's frames, so issue them on one stream or order them
 * with events.  rt_last_ray_counts / rt_last_render_ms refer to such a frame like to any other. */
int rt_render_device_noalloc(RtScene* scene, const RtRenderParams* params, void* d_out);

/* Size of that static array in bytes. */
int rt_noalloc_scratch_bytes(uint64_t* out_bytes);

/* Bytes one shard writes in rt_render_device per view (equal for every shard_index: the tile count is
 * rounded up to a multiple of shard_count so that gathers are uniform). */
int rt_shard_bytes(const RtScene* scene, const RtRenderParams* params, uint64_t* out_bytes);

/* After an all-gather of the shard buffers (rank-major, rt_shard_bytes each) writes the row-major
 * frame to `d_frame`.  Stitches rows as web_app/src/main/model.ts:115-129 stitches worker rows. */
int rt_assemble_shards(RtScene* scene, const RtRenderParams* params, const void* d_gathered, void* d_frame);

/* ---- one frame over several GPUs, driven by one process -----------------------------------------------------------
 * The reference's counterpart is its worker pool: N workers, each with its own copy of the Scene, render interleaved
 * rows into their own buffers and one owner stitches them (web_app/src/main/model.ts:93-129,
 * web_app/wasm/src/renderer.rs:68-76, web_app/src/worker/render_worker.ts:28,48-52).  rt_multi_create replicates the
 * scene on every device of `device_ids` (all different), owns two streams and one issuing host thread per device (the
 * reference's workers render side by side, render_worker.ts:71-76) and one RCCL communicator set (ncclCommInitAll;
 * librccl is only loaded for two or more distinct devices); rt_multi_render lets device i render the tiles t with
 * t % n_devices == i, moves the shard buffers with one ncclAllGather, assembles the frame on the first device and copies
 * it to `out_pixels` (host memory, the layout of rt_render; with rt_multi_set_views n_views frames back to back).
 * `stream` of `params` must be NULL, shard_index 0 and shard_count 0 or 1 (RT_ERR_INVALID_ARGUMENT otherwise).  The
 * bytes equal rt_render's for any number of devices.  A frame whose ray lists overflow on some device is rendered
 * again with longer lists, as rt_render does.  On any failure every device is waited for before the call returns.
 * rt_multi_render_views renders `n_views` cameras (matrices back to back, the screen size of the last
 * rt_multi_set_camera / set_views) in passes of `views_per_pass` (<= RT_MAX_VIEWS) and writes the frames back to back:
 * exchange, assembly and copy of pass k run on their own streams and buffers while pass k + 1 is traced.  Afterwards the
 * renderer is back at the views of the last rt_multi_set_views.
 * To rehearse the schedule on a one-GPU machine `device_ids` may name the same device n times (all entries equal): the
 * shards then share that GPU and the exchange is device-to-device copies instead of the collective. */
typedef struct RtMulti RtMulti;
int rt_multi_create(const RtSceneDesc* desc, const int* device_ids, int n_devices, RtMulti** out_renderer);
int rt_multi_set_camera(RtMulti* renderer, const float screen_to_world[16], uint32_t width, uint32_t height);
int rt_multi_set_views(RtMulti* renderer, const float* screen_to_world, uint32_t n_views, uint32_t width, uint32_t height);
int rt_multi_render(RtMulti* renderer, const RtRenderParams* params, uint8_t* out_pixels);
int rt_multi_render_views(RtMulti* renderer, const RtRenderParams* params, const float* screen_to_world, uint32_t n_views,
                          uint32_t views_per_pass, uint8_t* out_pixels);
/* ray counts of the last pass summed over the devices; host wall-clock milliseconds of the last rt_multi_render /
 * rt_multi_render_views call as a whole (shards, exchange, assembly and the copy to the host - not a device time) */
int rt_multi_last_ray_counts(RtMulti* renderer, uint64_t counters[RT_N_COUNTERS]);
int rt_multi_last_render_ms(RtMulti* renderer, float* out_ms);
void rt_multi_destroy(RtMulti* renderer);

/* Ray counts of the last rt_render / rt_render_device on this handle (waits for it to finish).
 * Fills RT_CTR_RAYS..RT_CTR_TRANSMITTED and RT_CTR_WALK_*.  Fails with RT_ERR_OUT_OF_RANGE if a ray list overflowed in any
 * rt_render_device frame since the previous call of this function (such a frame's output is incomplete); every
 * overflow is reported once. */
int rt_last_ray_counts(RtScene* scene, uint64_t counters[RT_N_COUNTERS]);

/* Device memory behind the per-depth work lists of this handle, and how full they were in the frame rendered last
 * (waits for it to finish): out[RT_LIST_USE_BYTES] = bytes held by the handle's per-frame buffers (colour sums, lists,
 * walk stacks); RT_LIST_USE_*_SLOTS = entries one recursion depth can hold of each kind of list - rays, walk items ((ray,
 * instance) pairs for the BVH walk), hit records; RT_LIST_USE_RAYS / _WALK_ITEMS / _HITS = entries the fullest depth of that
 * frame held.  A handle starts with lists of four rays per pixel and depth; every frame counted by this call, by
 * rt_last_ray_counts or by rt_render is remembered, and the next time the handle makes buffers for a frame of that size
 * they are half as long again as the fullest lists seen (buffers made on the first guess are given back then).  Overflows
 * are handled as before: the frame is reported, the lists double.  No counterpart in the reference, whose recursion
 * (raytracer.rs:65-135) keeps its rays on the call stack. */
enum { RT_LIST_USE_BYTES = 0, RT_LIST_USE_RAY_SLOTS, RT_LIST_USE_WALK_SLOTS, RT_LIST_USE_HIT_SLOTS,
       RT_LIST_USE_RAYS, RT_LIST_USE_WALK_ITEMS, RT_LIST_USE_HITS, RT_N_LIST_USE };
int rt_last_list_use(RtScene* scene, uint64_t out[RT_N_LIST_USE]);

/* Time of the last rt_render / rt_render_device on this handle between its first and last
 * enqueued command, measured with HIP events on the stream the kernels ran on. */
int rt_last_render_ms(RtScene* scene, float* out_ms);

/* The part of rt_last_render_ms spent in the launches of the recursion depths (everything between the
 * primary-ray kernel and the resolve kernel), measured the same way. */
int rt_last_trace_ms(RtScene* scene, float* out_ms);

/* The kernels launched per recursion depth: analytic primitives and instance culling, local rays and root
 * boxes of the surviving (ray, instance) pairs (intersect.rs:296-308), BVH walk (intersect.rs:228-269),
 * shading of the closest hits (raytracer.rs:65-135) together with the Phong sums of the depth above, whose
 * shadow rays are known by then (raytracer.rs:137-187). */
enum { RT_KERNEL_SETUP = 0, RT_KERNEL_TRAVERSE, RT_KERNEL_SHADE, RT_N_LEVEL_KERNELS };

/* Kernel timing mode (off by default): every launch of the per-depth kernels is bracketed by HIP
 * events on the launch stream.  No counterpart in the reference (its only timing is native_app's
 * "Rendering took", main.rs:38-40); exists so that a benchmark can report the duration of one kernel. */
int rt_set_kernel_timing(RtScene* scene, int enable);

/* Summed duration (ms) and number of launches of each per-depth kernel in the last frame rendered in
 * kernel timing mode. */
int rt_last_kernel_ms(RtScene* scene, float out_ms[RT_N_LEVEL_KERNELS], uint32_t out_launches[RT_N_LEVEL_KERNELS]);

void rt_scene_destroy(RtScene* scene);

/* Test aids.  The first sets how many rays a list of one recursion depth can hold (0 = the default,
 * four per pixel, or what the handle has measured).  When a list overflows, the kernels raise a flag: rt_render then enlarges the lists
 * and renders the frame again (the second call reports how often it had to for its last frame);
 * after rt_render_device the overflow is reported by rt_last_ray_counts, and the next frame gets
 * longer lists. */
int rt_debug_set_queue_capacity(RtScene* scene, uint32_t capacity);
int rt_debug_last_regrown(RtScene* scene, uint32_t* out_count);

/* Thread-local, static storage; never NULL. */
const char* rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
