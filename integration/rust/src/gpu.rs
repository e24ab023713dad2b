//! gpu.rs -- the binding a maintainer of addrianyy/path-tracer would add (`mod gpu;` in
//! src/main.rs) to hand the render call of src/main.rs:110-114 to libpt_amd.so.
//!
//! Three parts:
//!   1. the `extern "C"` block and `#[repr(C)]` structs of include/pt_amd.h -- every struct and every
//!      export, checked against the header by tests/test_host_logic.py::test_rust_binding_matches_header
//!      (field names, field types and function names are parsed out of both files);
//!   2. `pack()`: the walk over the reference's `BvhNode` tree that fills the plain arrays, with the
//!      small `describe` hooks the type-erased trait objects need (listed as a patch below);
//!   3. `GpuRenderer`, the drop-in for `ParallelRenderer` at the call site.
//! NOT compiled in this repository (there is no rustc in the build image): the same ABI is driven end to
//! end by the C++ host (host/main.cpp, tests/test_gpu_host.py) and by ctypes (path_tracer_amd/_abi.py).
#![allow(non_camel_case_types, dead_code)]
use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};
use std::sync::Arc;

// ---------------------------------------------------------------------------------------------
// 1. include/pt_amd.h
// ---------------------------------------------------------------------------------------------

#[repr(C)] #[derive(Copy, Clone)]
pub struct pt_sphere { pub cx: f32, pub cy: f32, pub cz: f32, pub radius: f32, pub material: i32 }

#[repr(C)] #[derive(Copy, Clone)]
pub struct pt_material { pub kind: i32, pub albedo: [f32; 3], pub param: f32, pub texture: i32 }

#[repr(C)] #[derive(Copy, Clone)]
pub struct pt_texture { pub width: i32, pub height: i32, pub rgb8: *const u8 }

#[repr(C)] #[derive(Copy, Clone)]
pub struct pt_camera {
    pub lower_left_corner: [f32; 3], pub origin: [f32; 3], pub horizontal: [f32; 3], pub vertical: [f32; 3],
    pub width: i32, pub height: i32,
}

#[repr(C)] #[derive(Copy, Clone)]
pub struct pt_bvh_node { pub bmin: [f32; 3], pub skip: i32, pub bmax: [f32; 3], pub prim: i32 }

#[repr(C)] #[derive(Copy, Clone)]
pub struct pt_device_info {
    pub name: [c_char; 64], pub arch: [c_char; 32],
    pub compute_units: i32, pub wavefront_size: i32, pub clock_khz: i32, pub memory_clock_khz: i32, pub l2_bytes: i32,
    pub hbm_bytes: i64,
}

#[repr(C)] #[derive(Copy, Clone)]
pub struct pt_primitive { pub kind: i32, pub material: i32, pub transform: i32, pub axis: i32, pub flip: i32, pub v: [f32; 5] }

#[repr(C)] #[derive(Copy, Clone)]
pub struct pt_transform { pub offset: [f32; 3], pub sin_theta: f32, pub cos_theta: f32 }

#[repr(C)] #[derive(Copy, Clone)]
pub struct pt_integrator { pub max_bounces: i32, pub background_kind: i32, pub background: [f32; 3], pub rr_start: i32 }

pub enum pt_context {}
pub enum pt_group {}

pub const PT_OK: c_int = 0;
pub const PT_ERR_INVALID: c_int = -1;
pub const PT_ERR_HIP: c_int = -2;
pub const PT_ERR_NO_SCENE: c_int = -3;
pub const PT_ERR_NO_DEVICE: c_int = -4;
pub const PT_ERR_LIMIT: c_int = -5;
pub const PT_ERR_CANCELLED: c_int = -6;
pub const PT_LAMBERTIAN: i32 = 0;
pub const PT_METAL: i32 = 1;
pub const PT_DIELECTRIC: i32 = 2;
pub const PT_DIFFUSE_LIGHT: i32 = 3;
pub const PT_PRIM_SPHERE: i32 = 0;
pub const PT_PRIM_RECT: i32 = 1;
pub const PT_EXACT: i32 = 0;
pub const PT_FAST: i32 = 1;
pub const PT_WALK_REFERENCE: i32 = 0;
pub const PT_WALK_VIEW: i32 = 1;

pub type pt_progress_fn = Option<extern "C" fn(user: *mut c_void, samples_done: i32, samples_total: i32) -> c_int>;

extern "C" {
    // context
    pub fn pt_create(device: i32, out: *mut *mut pt_context) -> c_int;
    pub fn pt_destroy(ctx: *mut pt_context);
    pub fn pt_last_error(ctx: *const pt_context) -> *const c_char;
    pub fn pt_query(ctx: *const pt_context, out: *mut pt_device_info) -> c_int;
    // scene
    pub fn pt_upload_scene(ctx: *mut pt_context, spheres: *const pt_sphere, n_spheres: i32,
                           materials: *const pt_material, n_materials: i32,
                           textures: *const pt_texture, n_textures: i32,
                           nodes: *const pt_bvh_node, n_nodes: i32) -> c_int;
    pub fn pt_set_camera(ctx: *mut pt_context, camera: *const pt_camera) -> c_int;
    // render
    pub fn pt_render(ctx: *mut pt_context, samples: i32, seed: u64, sample_begin: i32, sample_end: i32,
                     rgb8: *mut u8, radiance: *mut f32, hit_id: *mut i32) -> c_int;
    pub fn pt_render_device(ctx: *mut pt_context, samples: i32, seed: u64, sample_begin: i32, sample_end: i32,
                            accumulate: i32, d_radiance: *mut f32, stream: *mut c_void) -> c_int;
    pub fn pt_resolve_device(ctx: *mut pt_context, samples: i32, d_radiance: *const f32, d_rgb8: *mut u8,
                             stream: *mut c_void) -> c_int;
    pub fn pt_primary_hits_device(ctx: *mut pt_context, samples: i32, sample_index: i32, d_hit_id: *mut i32,
                                  d_hit_t: *mut f32, stream: *mut c_void) -> c_int;
    pub fn pt_render_device_static(ctx: *mut pt_context, samples: i32, seed: u64, sample_begin: i32, sample_end: i32,
                                   accumulate: i32, d_radiance: *mut f32, stream: *mut c_void) -> c_int;
    pub fn pt_render_progress(ctx: *mut pt_context, samples: i32, seed: u64, passes: i32,
                              progress: pt_progress_fn, user: *mut c_void,
                              rgb8: *mut u8, radiance: *mut f32) -> c_int;
    pub fn pt_last_path_kernel_ms(ctx: *mut pt_context) -> f32;
    // tuning
    pub fn pt_set_mode(ctx: *mut pt_context, mode: i32) -> c_int;
    pub fn pt_set_walk_order(ctx: *mut pt_context, order: i32) -> c_int;
    pub fn pt_get_walk_order(ctx: *const pt_context) -> c_int;
    pub fn pt_set_primary_lists(ctx: *mut pt_context, enabled: i32) -> c_int;
    pub fn pt_get_primary_lists(ctx: *const pt_context) -> c_int;
    pub fn pt_last_path_kernel_clock_ghz(ctx: *mut pt_context) -> f32;
    pub fn pt_clock_probe(ctx: *mut pt_context, ghz: *mut f32) -> c_int;
    pub fn pt_set_lanes_per_pixel(ctx: *mut pt_context, lanes: i32) -> c_int;
    pub fn pt_get_lanes_per_pixel(ctx: *const pt_context, samples: i32, sample_begin: i32, sample_end: i32) -> c_int;
    // start-up checks: each must report 0 mismatches
    pub fn pt_selftest_unary(ctx: *mut pt_context, func: i32, mismatches: *mut u64) -> c_int;
    pub fn pt_selftest_grid(ctx: *mut pt_context, samples: i32, mismatches: *mut u64) -> c_int;
    pub fn pt_selftest_division(ctx: *mut pt_context, log2_pairs: i32, mismatches: *mut u64) -> c_int;
    // extension (Cornell-box primitives, lights, integrator options)
    pub fn pt_upload_scene_ext(ctx: *mut pt_context, prims: *const pt_primitive, n_prims: i32,
                               materials: *const pt_material, n_materials: i32,
                               textures: *const pt_texture, n_textures: i32,
                               transforms: *const pt_transform, n_transforms: i32,
                               nodes: *const pt_bvh_node, n_nodes: i32) -> c_int;
    pub fn pt_set_integrator(ctx: *mut pt_context, integrator: *const pt_integrator) -> c_int;
    pub fn pt_bvh_build_ext(prims: *const pt_primitive, n_prims: i32, transforms: *const pt_transform,
                            n_transforms: i32, out: *mut pt_bvh_node, capacity: i32) -> i32;
    pub fn pt_gen_cornell_box(prims: *mut pt_primitive, prim_capacity: i32, materials: *mut pt_material,
                              material_capacity: i32, transforms: *mut pt_transform, transform_capacity: i32) -> i32;
    // several GPUs of one node
    pub fn pt_group_create(devices: *const i32, n_devices: i32, out: *mut *mut pt_group) -> c_int;
    pub fn pt_group_destroy(group: *mut pt_group);
    pub fn pt_group_last_error(group: *const pt_group) -> *const c_char;
    pub fn pt_group_size(group: *const pt_group) -> i32;
    pub fn pt_group_context(group: *mut pt_group, rank: i32) -> *mut pt_context;
    pub fn pt_group_upload_scene(group: *mut pt_group, spheres: *const pt_sphere, n_spheres: i32,
                                 materials: *const pt_material, n_materials: i32,
                                 textures: *const pt_texture, n_textures: i32,
                                 nodes: *const pt_bvh_node, n_nodes: i32) -> c_int;
    pub fn pt_group_set_camera(group: *mut pt_group, camera: *const pt_camera) -> c_int;
    pub fn pt_group_upload_scene_ext(group: *mut pt_group, prims: *const pt_primitive, n_prims: i32,
                                     materials: *const pt_material, n_materials: i32,
                                     textures: *const pt_texture, n_textures: i32,
                                     transforms: *const pt_transform, n_transforms: i32,
                                     nodes: *const pt_bvh_node, n_nodes: i32) -> c_int;
    pub fn pt_group_set_integrator(group: *mut pt_group, integrator: *const pt_integrator) -> c_int;
    pub fn pt_group_render(group: *mut pt_group, samples: i32, seed: u64, rgb8: *mut u8, radiance: *mut f32) -> c_int;
    pub fn pt_group_set_timeout(group: *mut pt_group, milliseconds: i32) -> c_int;
    pub fn pt_group_rccl_version(group: *const pt_group, compiled: *mut i32, loaded: *mut i32) -> c_int;
    pub fn pt_group_inject_fault(group: *mut pt_group, rank: i32, kind: i32) -> c_int;
    pub fn pt_group_shard(samples: i32, rank: i32, n: i32, begin: *mut i32, end: *mut i32);
    pub fn pt_group_timings(group: *const pt_group, kernel_ms: *mut f32, capacity: i32, reduce_ms: *mut f32) -> c_int;
    // host-side helpers (pure CPU)
    pub fn pt_bvh_build(spheres: *const pt_sphere, n_spheres: i32, out: *mut pt_bvh_node, capacity: i32) -> i32;
    pub fn pt_camera_look_at(eye: *const f32, target: *const f32, up: *const f32, fov_degrees: f32,
                             width: i32, height: i32, out: *mut pt_camera) -> c_int;
    pub fn pt_gen_simple_scene(spheres: *mut pt_sphere, materials: *mut pt_material) -> i32;
    pub fn pt_gen_random_scene(scene_seed: u64, a0: i32, a1: i32, b0: i32, b1: i32,
                               spheres: *mut pt_sphere, materials: *mut pt_material, capacity: i32) -> i32;
}

// ---------------------------------------------------------------------------------------------
// 2. pack(): the reference's objects -> the plain arrays
// ---------------------------------------------------------------------------------------------
//
// The reference's scene is a tree of type-erased trait objects (`Box<DynTraceable>` in `BvhNode::Leaf`,
// src/scene/bvh.rs:7-10; `Arc<dyn Material>`, src/material/mod.rs:15; `Arc<dyn Texture>`,
// src/texture/mod.rs:11) whose fields are private to their modules.  Each trait therefore gets ONE method
// that describes the object in the terms below, implemented next to the private fields.  The whole patch
// to the reference (method bodies only; nothing else changes):
//
//   src/traceable/mod.rs   trait Traceable        + fn describe(&self) -> gpu::TraceableDesc;
//   src/traceable/sphere.rs impl Traceable for Sphere
//                            + fn describe(&self) -> gpu::TraceableDesc {
//                            +     gpu::TraceableDesc::Sphere { center: self.center.extract_array(), radius: self.radius,
//                            +                                  material: self.material.clone() } }
//   src/material/mod.rs    trait Material         + fn describe(&self) -> gpu::MaterialDesc;
//   src/material/lambertian.rs impl Material for Lambertian
//                            + fn describe(&self) -> gpu::MaterialDesc { match &self.albedo {
//                            +     Albedo::Solid(c)   => gpu::MaterialDesc::Lambertian(gpu::AlbedoDesc::Solid(c.extract_array())),
//                            +     Albedo::Texture(t) => gpu::MaterialDesc::Lambertian(t.describe(t)) } }
//   src/material/metal.rs      + fn describe(&self) -> gpu::MaterialDesc {
//                            +     gpu::MaterialDesc::Metal { albedo: self.albedo.extract_array(), fuziness: self.fuziness } }
//   src/material/dielectric.rs + fn describe(&self) -> gpu::MaterialDesc { gpu::MaterialDesc::Dielectric { ref_idx: self.ref_idx } }
//   src/texture/mod.rs     trait Texture          + fn describe(&self, me: &SharedTexture) -> gpu::AlbedoDesc;
//   src/texture/solid.rs       + fn describe(&self, _: &SharedTexture) -> gpu::AlbedoDesc { gpu::AlbedoDesc::Solid(self.color.extract_array()) }
//   src/texture/picture.rs     + fn describe(&self, me: &SharedTexture) -> gpu::AlbedoDesc {
//                            +     gpu::AlbedoDesc::Picture { key: Arc::as_ptr(me) as *const () as usize, width: self.image.width() as i32,
//                            +                                height: self.image.height() as i32, rgb8: self.image.as_raw().as_ptr() } }
//   src/scene/mod.rs       impl Scene             + pub fn bvh_root(&self) -> Option<&BvhNode> { self.bvh_root.as_ref() }
//                          (and `mod bvh;` -> `pub(crate) mod bvh;` so that gpu.rs can name BvhNode)
//   src/raytracer.rs       impl Raytracer         + pub fn parts(&self) -> (&Scene, &Camera, usize) { (&self.scene, &self.camera, self.samples) }
//   src/math/camera.rs     impl Camera            + pub fn describe(&self) -> gpu::pt_camera { gpu::pt_camera {
//                            +     lower_left_corner: self.lower_left_corner.extract_array(), origin: self.origin.extract_array(),
//                            +     horizontal: self.horizontal.extract_array(), vertical: self.vertical.extract_array(),
//                            +     width: self.width as i32, height: self.height as i32 } }
//
// A SolidTexture under a Lambertian is folded into a solid albedo: Lambertian::scatter multiplies by
// `albedo.color(u, v, p)`, which for SolidTexture is the constant (src/texture/solid.rs:17-19).

use crate::material::SharedMaterial;
use crate::raytracer::Raytracer;
use crate::scene::bvh::BvhNode;

pub enum AlbedoDesc {
    Solid([f32; 3]),
    /// `key` identifies the shared texture object (its Arc address); `rgb8` stays owned by the texture,
    /// which the scene keeps alive for as long as the packed arrays are used
    Picture { key: usize, width: i32, height: i32, rgb8: *const u8 },
}
pub enum MaterialDesc {
    Lambertian(AlbedoDesc),
    Metal { albedo: [f32; 3], fuziness: f32 },
    Dielectric { ref_idx: f32 },
}
pub enum TraceableDesc {
    Sphere { center: [f32; 3], radius: f32, material: SharedMaterial },
}

pub struct PackedScene {
    pub spheres: Vec<pt_sphere>,
    pub materials: Vec<pt_material>,
    pub textures: Vec<pt_texture>,
    pub nodes: Vec<pt_bvh_node>,
    pub camera: pt_camera,
    pub samples: usize,
}

struct Packer {
    out: PackedScene,
    material_ids: HashMap<usize, i32>,   // Arc<dyn Material> address -> index in `materials`
    texture_ids: HashMap<usize, i32>,    // Arc<dyn Texture> address -> index in `textures`
}

impl Packer {
    /// One `pt_material` per distinct `Arc<dyn Material>` (src/material/mod.rs:15): the random scene shares
    /// nothing, `simple_scene` shares nothing either, but a user scene may hand one Arc to many spheres.
    fn material(&mut self, material: &SharedMaterial) -> i32 {
        let key = Arc::as_ptr(material) as *const () as usize;
        if let Some(&id) = self.material_ids.get(&key) {
            return id;
        }
        let pod = match material.describe() {
            MaterialDesc::Lambertian(AlbedoDesc::Solid(c)) => pt_material { kind: PT_LAMBERTIAN, albedo: c, param: 0.0, texture: -1 },
            MaterialDesc::Lambertian(AlbedoDesc::Picture { key, width, height, rgb8 }) => {
                let next = self.out.textures.len() as i32;
                let textures = &mut self.out.textures;
                let id = *self.texture_ids.entry(key).or_insert_with(|| {
                    textures.push(pt_texture { width, height, rgb8 });
                    next
                });
                pt_material { kind: PT_LAMBERTIAN, albedo: [0.0; 3], param: 0.0, texture: id }
            }
            MaterialDesc::Metal { albedo, fuziness } => pt_material { kind: PT_METAL, albedo, param: fuziness, texture: -1 },
            MaterialDesc::Dielectric { ref_idx } => pt_material { kind: PT_DIELECTRIC, albedo: [0.0; 3], param: ref_idx, texture: -1 },
        };
        let id = self.out.materials.len() as i32;
        self.out.materials.push(pod);
        self.material_ids.insert(key, id);
        id
    }

    /// DFS pre-order over the tree Scene::construct_bvh built (src/scene/bvh.rs:13-91), left child before
    /// right: node i's left child is i + 1, `skip` is the first node after its subtree -- the order in which
    /// BvhNode::trace visits (src/scene/bvh.rs:106-127).  Iterative: the reference's tree is as deep as its
    /// splits are lopsided.
    fn flatten(&mut self, root: &BvhNode) {
        enum Step<'a> { Enter(&'a BvhNode), Close(usize) }
        let mut stack = vec![Step::Enter(root)];
        while let Some(step) = stack.pop() {
            match step {
                Step::Close(me) => self.out.nodes[me].skip = self.out.nodes.len() as i32,
                Step::Enter(node) => {
                    let me = self.out.nodes.len();
                    let bbox = node.bounding_box();
                    self.out.nodes.push(pt_bvh_node { bmin: bbox.min.extract_array(), skip: 0, bmax: bbox.max.extract_array(), prim: -1 });
                    stack.push(Step::Close(me));
                    match node {
                        BvhNode::Leaf(_, object) => match object.describe() {
                            TraceableDesc::Sphere { center, radius, material } => {
                                let material = self.material(&material);
                                self.out.nodes[me].prim = self.out.spheres.len() as i32;
                                self.out.spheres.push(pt_sphere { cx: center[0], cy: center[1], cz: center[2], radius, material });
                            }
                        },
                        BvhNode::Split(_, children) => {
                            stack.push(Step::Enter(&children.1));   // popped second
                            stack.push(Step::Enter(&children.0));   // popped first: left before right
                        }
                    }
                }
            }
        }
    }
}

/// Everything `pt_upload_scene` + `pt_set_camera` need, from a constructed `Raytracer`
/// (src/raytracer.rs:36-44: the BVH exists once `Raytracer::new` has returned).  Primitive ids -- the hit-id
/// AOV -- follow the order of the leaves in the tree: `Scene::construct_bvh` moves the objects out of
/// `Scene::objects` (src/scene/mod.rs:58), so the Scene::add order is gone by then.
pub fn pack(raytracer: &Raytracer) -> PackedScene {
    let (scene, camera, samples) = raytracer.parts();
    let mut packer = Packer {
        out: PackedScene { spheres: Vec::new(), materials: Vec::new(), textures: Vec::new(), nodes: Vec::new(),
                           camera: camera.describe(), samples },
        material_ids: HashMap::new(),
        texture_ids: HashMap::new(),
    };
    packer.flatten(scene.bvh_root().expect("Raytracer::new constructs the BVH"));
    packer.out
}

// ---------------------------------------------------------------------------------------------
// 3. the renderer at the call site
// ---------------------------------------------------------------------------------------------

pub struct GpuRenderer { ctx: *mut pt_context }

impl GpuRenderer {
    /// replaces ParallelRenderer::new() (src/main.rs:97)
    pub fn new(device: i32) -> Result<Self, String> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { pt_create(device, &mut ctx) };
        if rc != PT_OK { return Err(last_error(std::ptr::null())); }
        Ok(Self { ctx })
    }

    pub fn upload(&mut self, s: &PackedScene) -> Result<(), String> {
        let rc = unsafe {
            pt_upload_scene(self.ctx, s.spheres.as_ptr(), s.spheres.len() as i32, s.materials.as_ptr(),
                            s.materials.len() as i32, s.textures.as_ptr(), s.textures.len() as i32,
                            s.nodes.as_ptr(), s.nodes.len() as i32)
        };
        if rc != PT_OK { return Err(last_error(self.ctx)); }
        let rc = unsafe { pt_set_camera(self.ctx, &s.camera) };
        if rc != PT_OK { return Err(last_error(self.ctx)); }
        Ok(())
    }

    /// replaces renderer.render(&context, &mut buffer, ..) (src/main.rs:110-114).  `buffer` is the
    /// Vec<[u8; 3]> of main.rs:98.  Afterwards the caller sets stats.pixels_done = pixel_count so the
    /// reporter thread (main.rs:63-66) terminates.
    pub fn render(&mut self, samples: usize, seed: u64, buffer: &mut [[u8; 3]]) -> Result<(), String> {
        let rc = unsafe {
            pt_render(self.ctx, samples as i32, seed, 0, (samples * samples) as i32,
                      buffer.as_mut_ptr() as *mut u8, std::ptr::null_mut(), std::ptr::null_mut())
        };
        if rc != PT_OK { Err(last_error(self.ctx)) } else { Ok(()) }
    }
}

impl Drop for GpuRenderer { fn drop(&mut self) { unsafe { pt_destroy(self.ctx) } } }

fn last_error(ctx: *const pt_context) -> String {
    unsafe { CStr::from_ptr(pt_last_error(ctx)).to_string_lossy().into_owned() }
}
