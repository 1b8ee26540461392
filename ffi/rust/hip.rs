//! `lib_raytracer/src/hip.rs` - the frame-level renderer of the crate on the MI355X library `librt_hip.so`.
//!
//! What a maintainer adds to the reference to use the HIP path: the `#[repr(C)]` mirrors of `include/rt_hip.h`, the
//! `extern "C"` block, `flatten(&Scene)` (the public fields of `scene.rs:6-20` into `RtSceneDesc`), `HipRenderer`
//! with the four methods of `GpuRenderer` (`web_app/wasm/src/gpu_renderer.rs:44-68, 280, 356-364`), and the bridge
//! from a `MeshLoader` (`scene_file/mesh_loader.rs:5-11`) to the library's `RtMeshLoadFn` callback.
//! Not compiled in this repository (its image has no Rust toolchain); `rust_raytracer_amd/abi.py` is the same mirror
//! in ctypes and is exercised by the tests.  Add `pub mod hip;` to `lib.rs` and `println!("cargo:rustc-link-lib=rt_hip")`
//! to a build script.
use std::ffi::{c_char, c_int, c_void, CStr};

use nalgebra_glm as glm;

use crate::raytracing::bvh::{Node, NodeType};
use crate::raytracing::{Background, Instance, Light, Material, MaterialType, Mesh, MeshTriangle, Plane, Sphere, Triangle};
use crate::Scene;

// ---- include/rt_hip.h, field for field -----------------------------------------------------------------------------
#[repr(C)] #[derive(Clone, Copy)] pub struct RtLight { pub position: [f32; 4], pub ambient: [f32; 3], pub diffuse: [f32; 3], pub specular: [f32; 3] }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtMaterial { pub emissive: [f32; 3], pub ambient: [f32; 3], pub diffuse: [f32; 3], pub specular: [f32; 3],
                                                          pub shininess: f32, pub material_type: u32, pub index_inner: f32, pub index_outer: f32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtPlane { pub normal: [f32; 3], pub distance: f32, pub material: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtSphere { pub center: [f32; 3], pub radius: f32, pub material: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtTriangle { pub vertices: [[f32; 3]; 3], pub normals: [[f32; 3]; 3], pub face_normal: [f32; 3], pub material: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtBvhNode { pub aabb_min: [f32; 3], pub aabb_max: [f32; 3], pub is_leaf: u32, pub child_left: u32,
                                                         pub child_right: u32, pub triangle_count: u32, pub triangle_indices: [u32; 5] }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtMesh { pub triangle_begin: u32, pub triangle_end: u32, pub node_begin: u32, pub node_end: u32, pub max_depth: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtMeshInstance { pub rotation_scale: [f32; 16], pub rotation_scale_inverse: [f32; 16], pub model: [f32; 16],
                                                              pub model_inverse: [f32; 16], pub mesh: u32, pub material_override: i32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtBackground { pub kind: u32, pub color: [f32; 3] }
#[repr(C)] pub struct RtSceneDesc {
    pub background: RtBackground,
    pub n_lights: u32, pub n_materials: u32, pub n_planes: u32, pub n_spheres: u32, pub n_triangles: u32,
    pub n_mesh_triangles: u32, pub n_mesh_bvh_nodes: u32, pub n_meshes: u32, pub n_mesh_instances: u32,
    pub lights: *const RtLight, pub materials: *const RtMaterial, pub planes: *const RtPlane, pub spheres: *const RtSphere,
    pub triangles: *const RtTriangle, pub mesh_triangles: *const RtTriangle, pub mesh_bvh_nodes: *const RtBvhNode,
    pub meshes: *const RtMesh, pub mesh_instances: *const RtMeshInstance,
}
#[repr(C)] pub struct RtRenderParams { pub max_depth: u32, pub output_format: u32, pub shard_index: u32, pub shard_count: u32,
                                       pub mode: u32, pub reserved: u32, pub stream: *mut c_void }
pub const RT_N_COUNTERS: usize = 17;
pub const RT_N_LIST_USE: usize = 7;
#[repr(C)] pub struct RtAux { pub prim_id: *mut u32, pub instance_id: *mut u32, pub flags: *mut u8, pub counters: [u64; RT_N_COUNTERS], pub hit_t: *mut f32 }
pub enum RtScene {}
pub enum RtMulti {}
pub const RT_FORMAT_RGB8: u32 = 0;
pub const RT_FORMAT_RGBA8: u32 = 1;
/// `RtMeshLoadFn` of include/rt_host.h
pub type RtMeshLoadFn = unsafe extern "C" fn(user: *mut c_void, file_name: *const c_char, out_data: *mut *const c_char, out_len: *mut usize,
                                             out_release: *mut Option<unsafe extern "C" fn(*mut c_void, *const c_char, usize)>) -> c_int;

#[link(name = "rt_hip")]
extern "C" {
    pub fn rt_scene_create(desc: *const RtSceneDesc, device_id: c_int, out: *mut *mut RtScene) -> c_int;
    pub fn rt_scene_set_camera(s: *mut RtScene, screen_to_world: *const f32, width: u32, height: u32) -> c_int;
    pub fn rt_scene_set_views(s: *mut RtScene, screen_to_world: *const f32, n_views: u32, width: u32, height: u32) -> c_int;
    pub fn rt_render(s: *mut RtScene, p: *const RtRenderParams, out_pixels: *mut u8, aux: *mut RtAux) -> c_int;
    pub fn rt_render_device(s: *mut RtScene, p: *const RtRenderParams, d_out: *mut c_void) -> c_int;
    pub fn rt_render_device_noalloc(s: *mut RtScene, p: *const RtRenderParams, d_out: *mut c_void) -> c_int;
    pub fn rt_last_ray_counts(s: *mut RtScene, counters: *mut u64) -> c_int;
    pub fn rt_last_list_use(s: *mut RtScene, out: *mut u64) -> c_int;
    pub fn rt_scene_destroy(s: *mut RtScene);
    pub fn rt_multi_create(desc: *const RtSceneDesc, device_ids: *const c_int, n_devices: c_int, out: *mut *mut RtMulti) -> c_int;
    pub fn rt_multi_set_camera(m: *mut RtMulti, screen_to_world: *const f32, width: u32, height: u32) -> c_int;
    pub fn rt_multi_render(m: *mut RtMulti, p: *const RtRenderParams, out_pixels: *mut u8) -> c_int;
    pub fn rt_multi_render_views(m: *mut RtMulti, p: *const RtRenderParams, screen_to_world: *const f32, n_views: u32, views_per_pass: u32, out_pixels: *mut u8) -> c_int;
    pub fn rt_multi_destroy(m: *mut RtMulti);
    pub fn rt_last_error() -> *const c_char;
}

fn check(rc: c_int) {
    // the reference's own error convention is panic!/unwrap (parser.rs:151-216, gpu_renderer.rs:47-49); nothing unwinds across the FFI
    if rc != 0 { panic!("rt_hip: {} (code {rc})", unsafe { CStr::from_ptr(rt_last_error()) }.to_string_lossy()) }
}

fn v3(v: &glm::Vec3) -> [f32; 3] { [v.x, v.y, v.z] }
/// nalgebra stores a `glm::Mat4` column-major: `as_slice()` is the 16 floats the C side expects
fn m16(m: &glm::Mat4) -> [f32; 16] { let mut out = [0.0; 16]; out.copy_from_slice(m.as_slice()); out }

fn flat_triangle(t: &Triangle) -> RtTriangle {
    RtTriangle { vertices: [v3(&t.vertices[0]), v3(&t.vertices[1]), v3(&t.vertices[2])],
                 normals: [v3(&t.normals[0]), v3(&t.normals[1]), v3(&t.normals[2])],
                 face_normal: v3(t.normal()), material: t.material.0 as u32 }            // geometry/triangle.rs:6-27
}

/// The arrays behind an `RtSceneDesc`; keep it alive for as long as the description is used (one `rt_scene_create` call).
pub struct FlatScene {
    lights: Vec<RtLight>, materials: Vec<RtMaterial>, planes: Vec<RtPlane>, spheres: Vec<RtSphere>, triangles: Vec<RtTriangle>,
    mesh_triangles: Vec<RtTriangle>, mesh_bvh_nodes: Vec<RtBvhNode>, meshes: Vec<RtMesh>, mesh_instances: Vec<RtMeshInstance>,
    background: RtBackground,
}

impl FlatScene {
    pub fn desc(&self) -> RtSceneDesc {
        RtSceneDesc {
            background: self.background,
            n_lights: self.lights.len() as u32, n_materials: self.materials.len() as u32, n_planes: self.planes.len() as u32,
            n_spheres: self.spheres.len() as u32, n_triangles: self.triangles.len() as u32,
            n_mesh_triangles: self.mesh_triangles.len() as u32, n_mesh_bvh_nodes: self.mesh_bvh_nodes.len() as u32,
            n_meshes: self.meshes.len() as u32, n_mesh_instances: self.mesh_instances.len() as u32,
            lights: self.lights.as_ptr(), materials: self.materials.as_ptr(), planes: self.planes.as_ptr(), spheres: self.spheres.as_ptr(),
            triangles: self.triangles.as_ptr(), mesh_triangles: self.mesh_triangles.as_ptr(), mesh_bvh_nodes: self.mesh_bvh_nodes.as_ptr(),
            meshes: self.meshes.as_ptr(), mesh_instances: self.mesh_instances.as_ptr(),
        }
    }
}

/// `Scene`'s public fields (scene.rs:6-20) as the plain arrays of include/rt_hip.h.  Material and mesh references are
/// already indices (parser.rs:124-144); node and triangle indices stay absolute, as the reference keeps them.
pub fn flatten(scene: &Scene) -> FlatScene {
    let lights = scene.lights.iter().map(|l: &Light| RtLight {                                             // light.rs:4-13
        position: [l.position.x, l.position.y, l.position.z, l.position.w],
        ambient: v3(&l.color.ambient), diffuse: v3(&l.color.diffuse), specular: v3(&l.color.specular) }).collect();
    let materials = scene.materials.iter().map(|m: &Material| {                                            // material.rs:6-25
        let (material_type, index_inner, index_outer) = match m.material_type {                           // numbering of gpu_types.rs:196-201
            MaterialType::Phong => (0, 0.0, 0.0),
            MaterialType::ReflectAndPhong => (1, 0.0, 0.0),
            MaterialType::ReflectAndRefract { index_inner, index_outer } => (2, index_inner, index_outer),
        };
        RtMaterial { emissive: v3(&m.emissive), ambient: v3(&m.ambient), diffuse: v3(&m.diffuse), specular: v3(&m.specular),
                     shininess: m.shininess, material_type, index_inner, index_outer }
    }).collect();
    let planes = scene.planes.iter().map(|p: &Plane| RtPlane { normal: v3(&p.normal), distance: p.distance, material: p.material.0 as u32 }).collect();
    let spheres = scene.spheres.iter().map(|s: &Sphere| RtSphere { center: v3(&s.center), radius: s.radius, material: s.material.0 as u32 }).collect();
    let triangles = scene.triangles.iter().map(flat_triangle).collect();
    let mesh_triangles = scene.mesh_triangles.iter().map(|t: &MeshTriangle| flat_triangle(&t.0)).collect();       // mesh_triangle.rs:3
    let mesh_bvh_nodes = scene.mesh_bvh_nodes.iter().map(|n: &Node| {                                      // bvh/node.rs:4-19
        let mut out = RtBvhNode { aabb_min: v3(&n.aabb.min), aabb_max: v3(&n.aabb.max), is_leaf: 0, child_left: 0, child_right: 0,
                                  triangle_count: 0, triangle_indices: [0; 5] };
        match &n.content {
            NodeType::Node { child_left, child_right } => { out.child_left = *child_left as u32; out.child_right = *child_right as u32; }
            NodeType::Leaf { triangle_indices } => {
                out.is_leaf = 1;
                out.triangle_count = triangle_indices.len() as u32;
                for (slot, index) in out.triangle_indices.iter_mut().zip(triangle_indices.iter()) { *slot = *index as u32; }
            }
        }
        out
    }).collect();
    let meshes = scene.meshes.iter().map(|m: &Mesh| RtMesh {                                               // geometry/mesh.rs:8-12, bvh/mod.rs:12-15
        triangle_begin: m.triangle_indices.start as u32, triangle_end: m.triangle_indices.end as u32,
        node_begin: m.bvh.bvh_node_indices.start as u32, node_end: m.bvh.bvh_node_indices.end as u32, max_depth: m.bvh.max_depth as u32 }).collect();
    let mesh_instances = scene.mesh_instances.iter().map(|i: &Instance<Mesh>| RtMeshInstance {            // instance.rs:6-14
        rotation_scale: m16(&i.rotation_scale), rotation_scale_inverse: m16(&i.rotation_scale_inverse),
        model: m16(&i.model), model_inverse: m16(&i.model_inverse),
        mesh: i.primitive_index as u32, material_override: i.material_override.map_or(-1, |m| m.0 as i32) }).collect();
    let background = match scene.background {                                                              // background.rs:3-7, gpu_types.rs:32-45
        Background::SolidColor(c) => RtBackground { kind: 0, color: v3(&c) },
        Background::ColoredDirection => RtBackground { kind: 1, color: [0.0; 3] },
    };
    FlatScene { lights, materials, planes, spheres, triangles, mesh_triangles, mesh_bvh_nodes, meshes, mesh_instances, background }
}

enum Handle { One(*mut RtScene), Several(*mut RtMulti) }

/// The crate's `GpuRenderer`, on the HIP library: owns the `Scene`, uploads it once, rewrites the camera per change.
pub struct HipRenderer { scene: Scene, handle: Handle, pub max_depth: u32 }

impl HipRenderer {
    /// `GpuRenderer::new` (gpu_renderer.rs:70-92) on one GPU
    pub fn new(width: usize, height: usize, scene: Scene) -> Self { Self::on_devices(width, height, scene, &[0]) }

    /// the same over several GPUs of this machine: tiles dealt round-robin, stitched on the first device (rt_multi_*)
    pub fn on_devices(width: usize, height: usize, mut scene: Scene, devices: &[c_int]) -> Self {
        scene.resize_screen(width, height);
        let flat = flatten(&scene);                                       // borrowed by the library for the call only
        let handle = if devices.len() == 1 {
            let mut h = std::ptr::null_mut();
            check(unsafe { rt_scene_create(&flat.desc(), devices[0], &mut h) });
            Handle::One(h)
        } else {
            let mut h = std::ptr::null_mut();
            check(unsafe { rt_multi_create(&flat.desc(), devices.as_ptr(), devices.len() as c_int, &mut h) });
            Handle::Several(h)
        };
        let r = HipRenderer { scene, handle, max_depth: 10 };             // MAX_RAY_RECURSION_DEPTH, raytracer.rs:9
        r.push_camera();
        r
    }

    fn push_camera(&self) {
        let dims = self.scene.camera().screen_dimensions;
        let m = self.scene.screen_to_world().as_slice().as_ptr();         // scene.rs:50-52, column-major
        match self.handle {
            Handle::One(h) => check(unsafe { rt_scene_set_camera(h, m, dims.x, dims.y) }),
            Handle::Several(h) => check(unsafe { rt_multi_set_camera(h, m, dims.x, dims.y) }),
        }
    }

    fn draw(&mut self, canvas: &mut [u8], output_format: u32, channels: usize) {
        let dims = self.scene.camera().screen_dimensions;
        assert_eq!(canvas.len(), dims.x as usize * dims.y as usize * channels);
        let p = RtRenderParams { max_depth: self.max_depth, output_format, shard_index: 0, shard_count: 0, mode: 0, reserved: 0,
                                 stream: std::ptr::null_mut() };
        match self.handle {
            Handle::One(h) => check(unsafe { rt_render(h, &p, canvas.as_mut_ptr(), std::ptr::null_mut()) }),
            Handle::Several(h) => check(unsafe { rt_multi_render(h, &p, canvas.as_mut_ptr()) }),
        }
    }

    /// `GpuRenderer::render(&mut self, canvas_u8: &mut [u8])` (gpu_renderer.rs:280-354): RGBA8, row 0 = top
    pub fn render(&mut self, canvas_u8: &mut [u8]) { self.draw(canvas_u8, RT_FORMAT_RGBA8, 4) }
    /// the layout of `Canvas` (canvas.rs:15-46): what `native_app::paint_scene` (main.rs:46-66) fills
    pub fn render_rgb(&mut self, canvas_u8: &mut [u8]) { self.draw(canvas_u8, RT_FORMAT_RGB8, 3) }
    /// gpu_renderer.rs:356-359
    pub fn resize_screen(&mut self, width: f32, height: f32) { self.scene.resize_screen(width as usize, height as usize); self.push_camera(); }
    /// gpu_renderer.rs:361-364
    pub fn turn_camera(&mut self, begin_x: f32, begin_y: f32, end_x: f32, end_y: f32) {
        self.scene.turn_camera(&glm::vec2(begin_x, begin_y), &glm::vec2(end_x, end_y));
        self.push_camera();
    }
}

impl Drop for HipRenderer {
    fn drop(&mut self) {
        match self.handle { Handle::One(h) => unsafe { rt_scene_destroy(h) }, Handle::Several(h) => unsafe { rt_multi_destroy(h) } }
    }
}

/// A mesh source for the library's own scene parser (`rt_host_scene_parse`, include/rt_host.h): file bytes by name,
/// which is what both of the reference's `MeshLoader`s read before they parse (`FilesystemMeshLoader`,
/// native_app/src/filesystem_mesh_loader.rs:16-30; `&AssetStore`, web_app/wasm/src/asset_store.rs).  Pass
/// `mesh_bytes_callback::<F>` as the `RtMeshLoadFn` and `&mut f as *mut F as *mut c_void` as its `user`.
pub unsafe extern "C" fn mesh_bytes_callback<F: FnMut(&str) -> Option<Vec<u8>>>(
    user: *mut c_void, file_name: *const c_char, out_data: *mut *const c_char, out_len: *mut usize,
    out_release: *mut Option<unsafe extern "C" fn(*mut c_void, *const c_char, usize)>) -> c_int {
    unsafe extern "C" fn release(_user: *mut c_void, data: *const c_char, len: usize) {
        drop(Vec::from_raw_parts(data as *mut u8, len, len));              // the boxed slice handed out below
    }
    let load = &mut *(user as *mut F);
    match CStr::from_ptr(file_name).to_str().ok().and_then(|name| load(name)) {
        Some(bytes) => {
            let boxed = bytes.into_boxed_slice();                         // capacity == length, as `release` assumes
            *out_len = boxed.len();
            *out_data = Box::into_raw(boxed) as *const c_char;
            *out_release = Some(release);
            0
        }
        None => 1,                                                        // the reference unwraps the io::Error (parser.rs:202-203)
    }
}
