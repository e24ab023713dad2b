//! gpu.rs -- the binding a maintainer of addrianyy/path-tracer would add (`mod gpu;` in
//! src/main.rs) to hand the render call of src/main.rs:110-114 to libpt_amd.so.
//! Mirrors include/pt_amd.h one to one.  NOT compiled here (no rustc in the build image).
#![allow(non_camel_case_types)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] #[derive(Copy, Clone)] pub struct pt_sphere { pub cx: f32, pub cy: f32, pub cz: f32, pub radius: f32, pub material: i32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct pt_material { pub kind: i32, pub albedo: [f32; 3], pub param: f32, pub texture: i32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct pt_texture { pub width: i32, pub height: i32, pub rgb8: *const u8 }
#[repr(C)] #[derive(Copy, Clone)] pub struct pt_camera {
    pub lower_left_corner: [f32; 3], pub origin: [f32; 3], pub horizontal: [f32; 3], pub vertical: [f32; 3],
    pub width: i32, pub height: i32,
}
#[repr(C)] #[derive(Copy, Clone)] pub struct pt_bvh_node { pub bmin: [f32; 3], pub skip: i32, pub bmax: [f32; 3], pub prim: i32 }
pub enum pt_context {}

pub const PT_LAMBERTIAN: i32 = 0;
pub const PT_METAL: i32 = 1;
pub const PT_DIELECTRIC: i32 = 2;

extern "C" {
    pub fn pt_create(device: i32, out: *mut *mut pt_context) -> c_int;
    pub fn pt_destroy(ctx: *mut pt_context);
    pub fn pt_last_error(ctx: *const pt_context) -> *const c_char;
    pub fn pt_upload_scene(ctx: *mut pt_context, spheres: *const pt_sphere, n_spheres: i32,
                           materials: *const pt_material, n_materials: i32,
                           textures: *const pt_texture, n_textures: i32,
                           nodes: *const pt_bvh_node, n_nodes: i32) -> c_int;
    pub fn pt_set_camera(ctx: *mut pt_context, camera: *const pt_camera) -> c_int;
    pub fn pt_render(ctx: *mut pt_context, samples: i32, seed: u64, sample_begin: i32, sample_end: i32,
                     rgb8: *mut u8, radiance: *mut f32, hit_id: *mut i32) -> c_int;
    pub fn pt_render_device(ctx: *mut pt_context, samples: i32, seed: u64, sample_begin: i32, sample_end: i32,
                            accumulate: i32, d_radiance: *mut f32, stream: *mut c_void) -> c_int;
    pub fn pt_resolve_device(ctx: *mut pt_context, samples: i32, d_radiance: *const f32, d_rgb8: *mut u8,
                             stream: *mut c_void) -> c_int;
    pub fn pt_render_progress(ctx: *mut pt_context, samples: i32, seed: u64, passes: i32,
                              progress: Option<extern "C" fn(*mut c_void, i32, i32) -> c_int>, user: *mut c_void,
                              rgb8: *mut u8, radiance: *mut f32) -> c_int;
    // optional start-up checks (include/pt_amd.h): both must report 0 mismatches
    pub fn pt_selftest_unary(ctx: *mut pt_context, func: i32, mismatches: *mut u64) -> c_int;
    pub fn pt_selftest_grid(ctx: *mut pt_context, samples: i32, mismatches: *mut u64) -> c_int;
    pub fn pt_selftest_division(ctx: *mut pt_context, log2_pairs: i32, mismatches: *mut u64) -> c_int;
}

/// What each `impl Material` / `impl Texture` / `impl Traceable` adds so that the type-erased
/// `Arc<dyn Material>` (src/material/mod.rs:15) can describe itself -- one method per trait:
///   trait Material { fn scatter(..); fn describe(&self, textures: &mut TexturePacker) -> pt_material; }
///   trait Traceable { fn trace(..); fn bounding_box(..); fn describe(&self) -> (Vec3 /*center*/, f32 /*radius*/, &SharedMaterial); }
/// Lambertian -> {PT_LAMBERTIAN, albedo or texture index}, Metal -> {PT_METAL, albedo, fuziness},
/// Dielectric -> {PT_DIELECTRIC, ref_idx}; PictureTexture hands out `image.as_raw()` (RGB8).
pub struct PackedScene {
    pub spheres: Vec<pt_sphere>, pub materials: Vec<pt_material>, pub textures: Vec<pt_texture>,
    pub nodes: Vec<pt_bvh_node>,
}

// Walks the BvhNode tree built by Scene::construct_bvh (src/scene/bvh.rs:7-10) in DFS pre-order,
// left before right, so the GPU walks exactly the tree the CPU renderer would.  Shared materials
// are de-duplicated by Arc pointer.  (Pseudo-Rust: needs the `describe` hooks above.)
//
// fn flatten(node: &BvhNode, out: &mut PackedScene, mats: &mut HashMap<*const (), i32>) {
//     let me = out.nodes.len();
//     let b = node.bounding_box();
//     out.nodes.push(pt_bvh_node { bmin: b.min.extract_array(), skip: 0, bmax: b.max.extract_array(), prim: -1 });
//     match node {
//         BvhNode::Leaf(_, obj) => {
//             let (c, r, m) = obj.describe();
//             let mi = *mats.entry(Arc::as_ptr(m) as *const ()).or_insert_with(|| { out.materials.push(m.describe(..)); .. });
//             out.nodes[me].prim = out.spheres.len() as i32;
//             out.spheres.push(pt_sphere { cx: c.0, cy: c.1, cz: c.2, radius: r, material: mi });
//         }
//         BvhNode::Split(_, kids) => { flatten(&kids.0, out, mats); flatten(&kids.1, out, mats); }
//     }
//     out.nodes[me].skip = out.nodes.len() as i32;
// }
// (primitive ids then follow DFS leaf order instead of Scene::add order; pass the insertion index
//  through `describe` if the hit-id AOV must match Scene::add order.)

pub struct GpuRenderer { ctx: *mut pt_context }

impl GpuRenderer {
    /// replaces ParallelRenderer::new() (src/main.rs:97)
    pub fn new(device: i32) -> Result<Self, String> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { pt_create(device, &mut ctx) };
        if rc != 0 { return Err(last_error(std::ptr::null())); }
        Ok(Self { ctx })
    }

    pub fn upload(&mut self, s: &PackedScene, camera: &pt_camera) -> Result<(), String> {
        let rc = unsafe {
            pt_upload_scene(self.ctx, s.spheres.as_ptr(), s.spheres.len() as i32, s.materials.as_ptr(),
                            s.materials.len() as i32, s.textures.as_ptr(), s.textures.len() as i32,
                            s.nodes.as_ptr(), s.nodes.len() as i32)
        };
        if rc != 0 { return Err(last_error(self.ctx)); }
        let rc = unsafe { pt_set_camera(self.ctx, camera) };
        if rc != 0 { return Err(last_error(self.ctx)); }
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
        if rc != 0 { Err(last_error(self.ctx)) } else { Ok(()) }
    }
}

impl Drop for GpuRenderer { fn drop(&mut self) { unsafe { pt_destroy(self.ctx) } } }

fn last_error(ctx: *const pt_context) -> String {
    unsafe { CStr::from_ptr(pt_last_error(ctx)).to_string_lossy().into_owned() }
}
