/*
 * pt_amd.h -- C ABI of libpt_amd.so, the MI355X (gfx950) replacement for the
 * radiance hot path of addrianyy/path-tracer.
 *
 * The reference has no FFI today.  The seam this ABI replaces is the one call
 *     renderer.render(&context, &mut buffer, |ctx, rng, start, pixels|
 *                         raytracer.render_fragment(start, pixels, &stats, rng))
 * at src/main.rs:110-114, i.e. ParallelRenderer::render
 * (src/parallel_renderer/mod.rs:155-203) driving Raytracer::render_fragment
 * (src/raytracer.rs:98-123) and everything below it.  The Rust host keeps
 * main.rs, the scene/material/texture builder API and image output; it packs
 * its objects into the plain-old-data arrays below (INTEGRATION.md shows the
 * `extern "C"` block and build.rs a maintainer would add).
 *
 * Conventions: every function returns PT_OK (0) or a negative pt_status; no
 * exception or unwinding ever crosses this boundary; pt_last_error() returns a
 * message for the most recent failure on that context.  The caller owns every
 * pointer it passes; the library copies at upload time and keeps no host
 * pointer after a call returns.  A context is used from one host thread at a
 * time.  Pointers named d_* are DEVICE pointers on the context's GPU; all
 * others are host pointers.
 */
#ifndef PT_AMD_H
#define PT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_context pt_context;

typedef enum {
    PT_OK = 0,
    PT_ERR_INVALID = -1,   /* bad argument (message says which)            */
    PT_ERR_HIP = -2,       /* a HIP runtime call failed                    */
    PT_ERR_NO_SCENE = -3,  /* render called before scene / camera upload   */
    PT_ERR_NO_DEVICE = -4, /* no usable gfx950 device                      */
    PT_ERR_LIMIT = -5,     /* size beyond a documented maximum             */
    PT_ERR_CANCELLED = -6  /* a progress callback asked to stop            */
} pt_status;

/* Sphere {center, radius, material}: src/traceable/sphere.rs:20-24.  `material`
 * indexes the materials array (shared Arc<dyn Material>s de-duplicated by the
 * host).  Primitive id (hit-id AOV) = the caller's index in this array, whatever
 * order the caller filled it in: Scene::add order (src/scene/mod.rs:46-48) for a
 * host that packs Scene::objects before construct_bvh drains it (the C++ and Python
 * hosts here), leaf order of the built tree for the Rust packer, which can only walk
 * the BvhNode tree afterwards (integration/rust/src/gpu.rs flatten). */
typedef struct {
    float cx, cy, cz, radius;
    int32_t material;
} pt_sphere;

/* src/material/{lambertian,metal,dielectric}.rs */
enum { PT_LAMBERTIAN = 0, PT_METAL = 1, PT_DIELECTRIC = 2 };
typedef struct {
    int32_t kind;     /* PT_LAMBERTIAN / PT_METAL / PT_DIELECTRIC                      */
    float albedo[3];  /* Lambertian::new_solid colour, Metal albedo; unused otherwise  */
    float param;      /* Metal: fuziness (metal.rs:9); Dielectric: ref_idx             */
    int32_t texture;  /* Lambertian::new(PictureTexture): texture index; else -1       */
} pt_material;

/* PictureTexture after image::open(..).into_rgb() (src/texture/picture.rs:9-13):
 * decoded RGB8, row-major from the top row.  JPEG decoding stays on the host. */
typedef struct {
    int32_t width, height;
    const uint8_t *rgb8;
} pt_texture;

/* The DERIVED fields of Camera (src/math/camera.rs:4-11).  Camera::new itself
 * stays on the host (it uses tan and the rsqrt normalise, camera.rs:14-34);
 * pt_camera_look_at below is a convenience that restates it. */
typedef struct {
    float lower_left_corner[3], origin[3], horizontal[3], vertical[3];
    int32_t width, height;
} pt_camera;

/* One node of the flattened BvhNode tree (src/scene/bvh.rs:7-10) in DFS
 * pre-order, left subtree before right.  skip = index of the first node after
 * this node's subtree (node_count for the root).  prim = sphere index for a
 * Leaf, -1 for a Split.  Bounds are the node's AABB (a Leaf's are never tested,
 * bvh.rs:109). */
typedef struct {
    float bmin[3];
    int32_t skip;
    float bmax[3];
    int32_t prim;
} pt_bvh_node;

/* ---- context ------------------------------------------------------------ */

/* Replaces ParallelRenderer::new (src/parallel_renderer/mod.rs:73-153): instead of
 * a CPU worker pool, binds HIP device `device` and creates the render stream. */
int pt_create(int32_t device, pt_context **out);
void pt_destroy(pt_context *ctx);
const char *pt_last_error(const pt_context *ctx);   /* ctx may be NULL: create failures */

/* Device facts for roofline reporting: name, CU count, clocks. */
typedef struct {
    char name[64];
    char arch[32];
    int32_t compute_units, wavefront_size, clock_khz, memory_clock_khz, l2_bytes;
    int64_t hbm_bytes;
} pt_device_info;
int pt_query(const pt_context *ctx, pt_device_info *out);

/* ---- scene -------------------------------------------------------------- */

/* Replaces Raytracer::new (src/raytracer.rs:36-44): takes ownership of a copy of
 * the scene and gets the BVH ready.  nodes == NULL / n_nodes == 0: the library
 * builds the tree with the algorithm of src/scene/bvh.rs:13-91 (pt_bvh_build). */
int pt_upload_scene(pt_context *ctx, const pt_sphere *spheres, int32_t n_spheres,
                    const pt_material *materials, int32_t n_materials,
                    const pt_texture *textures, int32_t n_textures,
                    const pt_bvh_node *nodes, int32_t n_nodes);
int pt_set_camera(pt_context *ctx, const pt_camera *camera);

/* ---- render ------------------------------------------------------------- */

/* Replaces renderer.render(..) at src/main.rs:110-114.  Blocking.  Renders sample
 * indices s in [sample_begin, sample_end) of the samples x samples grid
 * (s = sx * samples + sy, src/raytracer.rs:77-81); the full image is
 * [0, samples*samples).  Per-(pixel,sample) RNG streams derive from `seed`.
 * Outputs are host buffers, any may be NULL:
 *   rgb8        W*H*3 bytes, row-major, top row first -- the `buffer` of main.rs:98,
 *               resolved as sqrt(sum / samples^2) * 255 -> u8 (raytracer.rs:92-93,119-121).
 *               Only meaningful when the range is the full grid.
 *   radiance    W*H*3 fp32: the un-normalised radiance SUM over the range (AOV).
 *   hit_id      W*H int32: primitive id hit by the primary ray of sample index 0,
 *               -1 = miss (AOV; HitRecord has no id in the reference,
 *               src/traceable/mod.rs:11-17 -- id = Scene::add order). */
int pt_render(pt_context *ctx, int32_t samples, uint64_t seed, int32_t sample_begin, int32_t sample_end,
              uint8_t *rgb8, float *radiance, int32_t *hit_id);

/* Device-resident halves of pt_render, for callers that keep the radiance on the
 * GPU (spp-sharded multi-GPU: render a sample range per GPU, reduce the fp32
 * buffers with RCCL, resolve on the root).  Asynchronous on `stream` (a
 * hipStream_t; NULL = the context's own stream).
 *   pt_render_device : d_radiance[W*H*3] = (accumulate ? old : 0) + sum over the range
 *   pt_resolve_device: d_rgb8[W*H*3] from d_radiance (raytracer.rs:92-93,119-121)
 *   pt_primary_hits_device: d_hit_id[W*H] (and optional d_hit_t[W*H]) for sample index s
 * The launches of one context share its work-distribution scratch: issue them on ONE stream
 * at a time (back-to-back launches on the same stream are ordered and fine).  pt_render_device on a
 * second stream while the context's previous launch is still running returns PT_ERR_INVALID
 * instead of racing.  A dependency wait that gives up inside the path kernel (never observed) is
 * reported by pt_last_path_kernel_ms (-2) and by pt_render / pt_render_progress (PT_ERR_HIP),
 * not by these asynchronous calls. */
int pt_render_device(pt_context *ctx, int32_t samples, uint64_t seed, int32_t sample_begin,
                     int32_t sample_end, int32_t accumulate, float *d_radiance, void *stream);
int pt_resolve_device(pt_context *ctx, int32_t samples, const float *d_radiance, uint8_t *d_rgb8, void *stream);
int pt_primary_hits_device(pt_context *ctx, int32_t samples, int32_t sample_index, int32_t *d_hit_id,
                           float *d_hit_t, void *stream);

/* pt_render_device for callers that cannot allocate.  No counterpart in the reference: it is for
 * code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same
 * arguments, same launch, same results bit for bit; the only difference is where the launch's
 * work-distribution scratch (ticket counter, error word, one progress word per 8x8 tile) lives:
 * in a file-scope __device__ array of the library (1 MiB + 256 B on every GPU the library runs on,
 * held until the process ends) instead of the buffer pt_set_camera allocates.  The call itself
 * performs no hipMalloc / hipFree / allocating library call.
 *   limit : frames of at most 262,144 8x8 tiles (16.7 M pixels; 3840x2160 has 129,600); larger
 *           frames return PT_ERR_LIMIT.
 *   sharing: ALL contexts on one GPU share that GPU's array (it is cleared at the start of every
 *           launch), so two launches of this entry point must not overlap in time on one GPU.  The
 *           library keeps the array's in-flight record per GPU: a launch on another stream -- of
 *           this context or of any other context of the GPU -- while the array's previous launch has
 *           not completed returns PT_ERR_INVALID instead of racing (launches on ONE stream are
 *           ordered and fine).  GPUs beyond the 64th of a process are not tracked.
 *   capture: on a stream under HIP graph capture the call records ONE kernel node and nothing else (no
 *           memset: a launch leaves its work-distribution words at zero for the next one; no events, no
 *           in-flight check: ordering is the graph's business; the occupancy and attribute queries of the
 *           launch are cached from the first un-captured launch of the same scene size).
 *           pt_last_path_kernel_ms returns -1 for such a launch.  tests/test_zz_no_alloc.py captures
 *           pt_render_device_static + pt_resolve_device, replays the graph twice and compares with
 *           pt_render bit for bit. */
int pt_render_device_static(pt_context *ctx, int32_t samples, uint64_t seed, int32_t sample_begin,
                            int32_t sample_end, int32_t accumulate, float *d_radiance, void *stream);

/* pt_render in `passes` launches over consecutive sample-index ranges, calling
 * progress(user, samples_done, samples_total) on the calling thread after each one: the
 * counterpart of Statistics::pixels_done + the reporter thread (src/raytracer.rs:108-116,
 * src/main.rs:41-71), which count finished pixels where this counts finished samples of every
 * pixel.  Each pass continues the per-pixel sums of the previous one in sample order, so the
 * image equals pt_render's bit for bit whenever one lane owns a pixel (frames of >= 524,288
 * pixels, or pt_set_lanes_per_pixel(1)); small frames differ by summation order only.  A
 * non-zero return from the callback stops the render: PT_ERR_CANCELLED, outputs untouched. */
typedef int (*pt_progress_fn)(void *user, int32_t samples_done, int32_t samples_total);
int pt_render_progress(pt_context *ctx, int32_t samples, uint64_t seed, int32_t passes, pt_progress_fn progress,
                       void *user, uint8_t *rgb8, float *radiance);

/* Milliseconds the path kernel of the most recent pt_render / pt_render_device
 * took, measured with HIP events on the stream it was launched on.  Synchronises
 * that stream.  -1 if no render has run; -2 if that launch reported an error from inside the
 * kernel (pt_last_error says which) -- the way to check a pt_render_device launch. */
float pt_last_path_kernel_ms(pt_context *ctx);

/* Arithmetic mode of the path kernel.  No counterpart in the reference.
 *   PT_EXACT (default): every + - * / sqrt is one correctly rounded fp32 operation in the reference's
 *            order, nothing fused (Rust never contracts a * b + c): radiance equals the oracle's bit for bit.
 *   PT_FAST : opt-in.  Fused multiply-adds, hardware reciprocal / square-root estimates, the slab test
 *            with one fma per plane.  Not bit-equal to the EXACT result; within BASELINE's tolerance of
 *            it (>= 99.9 % of pixels within 1/255 after tonemapping at equal spp and seed, converged
 *            RMSE < 1e-3) on the 256-, 1024- and 4096-spp configurations (99.94 - 99.99 % measured).
 *            EXCEPTION: at 100 spp (BASELINE config 1, 400x225) it measures 99.78 %, BELOW the 99.9 %
 *            bar -- one path that takes another way moves a pixel by more than 1/255 at so few samples --
 *            so PT_FAST does not meet the north-star tolerance there (tests/test_gpu_fast.py holds it to
 *            99.5 % for that frame and says so).  Applies to pt_upload_scene scenes; the hit-id AOV, the
 *            resolve and the extension's scenes (pt_upload_scene_ext) always run EXACT. */
enum { PT_EXACT = 0, PT_FAST = 1 };
int pt_set_mode(pt_context *ctx, int32_t mode);

/* Order in which the walk visits a Split's two children.  No counterpart in the reference, whose
 * BvhNode::trace always enters the left child first (src/scene/bvh.rs:111-118).  The tree -- its boxes
 * and leaves, caller-supplied or built here -- is never changed; only the threading of the device copy is.
 *   PT_WALK_VIEW (default): the child whose box centre is nearer to the camera's origin first.  Primary
 *            rays then find their closest hit early and the boxes behind it fail their test against it:
 *            11 % fewer box tests and 16 % fewer sphere tests per path on the bench scene.  The node
 *            arrays are re-threaded (host, O(nodes)) by pt_set_camera and pt_upload_scene.
 *   PT_WALK_REFERENCE: left child first, the reference's own visiting order.
 * A ray's closest hit does not depend on the order in which its candidates are found, so both orders
 * return the same hit ids and the same radiance bit for bit (every parity test runs under PT_WALK_VIEW;
 * tests/test_gpu_parity.py::test_walk_orders_agree compares the two).  Two fine points: (1) candidates with
 * the SAME t -- two spheres of equal centre and squared radius, or a ray through the very circle where two
 * overlapping spheres meet (seen about once in 10^8 primary rays on the bench scene) -- where the reference
 * keeps the first one its walk meets (strict '<', src/traceable/sphere.rs:57): the device's leaves carry
 * their rank in the reference's order and the lower rank wins a tie, which restores the reference's choice
 * under any order; pt_upload_scene_ext scenes (coplanar rects tie by construction) are always walked in the
 * reference's order, and so are trees in which some box does not contain what is below it (the reference
 * builds such boxes for spheres of negative radius, src/traceable/sphere.rs:71-76: there a box test can hide
 * a closer hit and the order decides) -- pt_get_walk_order reports the order in effect; (2) a root and a box-entry distance
 * that agree to within an fp32 rounding error while a second hit lies within the same error of them --
 * needs a hit on the very point where a sphere touches its box together with a near-tie; never observed
 * (DESIGN.md section 11). */
enum { PT_WALK_REFERENCE = 0, PT_WALK_VIEW = 1 };
int pt_set_walk_order(pt_context *ctx, int32_t order);
int pt_get_walk_order(const pt_context *ctx);   /* the order in effect for the uploaded scene */

/* Primary-ray candidate lists.  No counterpart in the reference, whose every ray walks the tree from the root
 * (src/scene/mod.rs:24-34 -> src/scene/bvh.rs:106-127).  For every 8x8 pixel block of the camera's frame the library
 * lists the spheres a camera ray through the block can possibly hit (2.2 on average on the bench scenes): those whose
 * ball, inflated by a margin a hundred times the fp32 rounding of the kernel's own rays and tests, meets the block's
 * cone of directions (src/raytracer.rs:77-85, src/math/camera.rs:37-44).  A regular primary ray then makes, per
 * candidate, the box test of the Leaf's PARENT Split and, if it passes, Sphere::trace -- in rank order, with the same
 * tie rule -- instead of walking the tree: about 19 box tests per path fewer on the bench scene.  The parent's box
 * decides what the reference's walk decides with all the boxes on its way: boxes are nested and the slab test
 * (src/math/aabb.rs:25-36) is monotone in the box for a ray without NaNs, so a ray that passes the parent's test passes
 * every ancestor's.  (The box test cannot be dropped: where a sphere touches its boxes and a ray grazes it there, the slab
 * test is exact and Sphere::trace's discriminant is not, so the sphere test "hits" for rays the reference's boxes never
 * let through -- main.rs's eye sits at the height of the big spheres' tops, and the full BASELINE cfg 5 frame differed
 * in 11 pixels without it.)  Hits and radiance are the plain walk's bit for bit; what differs is only the
 * closest hit a box is tested against at a given moment, as with PT_WALK_VIEW's fine point (2) below.  Lists are built
 * only for trees whose boxes are nested (as PT_WALK_VIEW), never for pt_upload_scene_ext scenes or trees of fewer than
 * 32 nodes; a block with more than 8 candidates, and any ray with a zero or non-finite direction component, walks the
 * tree.  The hit-id AOV always walks the tree.  Rebuilt (a GPU kernel, then a stream sync) by pt_set_camera /
 * pt_upload_scene / pt_set_walk_order.  enabled = 0 turns the lists off (every ray walks the tree); default on.
 * pt_get_primary_lists: 1 if the uploaded scene + camera render with lists, 0 if not. */
int pt_set_primary_lists(pt_context *ctx, int32_t enabled);
int pt_get_primary_lists(const pt_context *ctx);

/* The shader clock of the context's GPU right now, in GHz: a two-millisecond chip-filling chain of fp32
 * multiplies (its own small kernel, 8 waves per SIMD like the path kernel) stamped with the shader-cycle
 * counter and the constant 100 MHz counter; the median over its workgroups of cycles per tick.  No
 * counterpart in the reference.  For measurement records: a frame's time is cycles / clock, and the
 * clock is the chip's choice under load -- bench.py calls this right before and right after its timed
 * region so that a slower box can be told from a slower kernel.  Blocking; allocates. */
int pt_clock_probe(pt_context *ctx, float *ghz);

/* The shader clock the most recent persistent path-kernel launch of the context ran at, in GHz: its first workgroup
 * stamps the shader-cycle counter and the constant 100 MHz counter once at its start and once at its end (two stamps per
 * launch, none inside the work).  Synchronises the launch's stream.  -1: no launch yet, a captured launch, or the
 * small-tree launch (scenes of fewer than 32 nodes), which does not stamp.  No counterpart in the reference. */
float pt_last_path_kernel_clock_ghz(pt_context *ctx);

/* How a pixel's samples are spread over GPU lanes.  0 (default) = automatic:
 * one lane owns all samples of a pixel whenever the frame alone fills the GPU,
 * which reproduces the reference's sequential fp32 sum order (raytracer.rs:88)
 * bit for bit; small frames split each pixel's range into `lanes` contiguous
 * chunks summed in order.  n > 0 forces n lanes per pixel. */
int pt_set_lanes_per_pixel(pt_context *ctx, int32_t lanes);
int pt_get_lanes_per_pixel(const pt_context *ctx, int32_t samples, int32_t sample_begin, int32_t sample_end);

/* Self-test, no counterpart in the reference: runs the library's short forms of 1/x (fn 0),
 * sqrt(x) (fn 1) and 1/sqrt(x) (fn 2) -- the three unary fp32 operations of vec.rs:38-44,95-97,
 * scene/mod.rs:31, sphere.rs:54 -- over ALL 2^32 fp32 inputs on the GPU against the correctly
 * rounded IEEE operation and returns the number of inputs whose result differs (NaN payloads
 * aside).  The library relies on this being 0. */
int pt_selftest_unary(pt_context *ctx, int32_t fn, uint64_t *mismatches);
/* Self-test, no counterpart in the reference: the sample-grid coordinates u, v of
 * raytracer.rs:77-85 for EVERY (pixel column, sx) and (pixel row, sy) of the current camera at
 * `samples`, computed the way the path kernel does (reciprocal + fused corrections) against the
 * IEEE divisions of the reference; returns the number that differ.  The library relies on 0. */
int pt_selftest_grid(pt_context *ctx, int32_t samples, uint64_t *mismatches);
/* Self-test, no counterpart in the reference: the division x / a of sphere.rs:55,62 the way the
 * path kernel does it for REGULAR rays (a's correctly rounded reciprocal, taken once per walk, and
 * two fused corrections) against the IEEE division, on 2^log2_pairs operand pairs (16..40): a over
 * the class's whole range [2^-32, 2^32], half of the numerators random below 2^53, half built so
 * that the quotient lies within two ulps of a rounding midpoint; then half as many pairs again with a's
 * mantissa pinned to all ones (a = dot(d, d) of a normalised direction).  Not exhaustive (2^64 pairs
 * exist); the proof is Markstein's theorem, this is its check.  Returns the number that differ. */
int pt_selftest_division(pt_context *ctx, int32_t log2_pairs, uint64_t *mismatches);

/* ---- extension: Cornell-box primitives, lights, integrator options ----------------
 * NOT in the reference: it has one primitive (Sphere, src/traceable/mod.rs:1-7), three
 * materials with `scatter` only (src/material/mod.rs:17-19), a fixed 5-bounce cap and the sky
 * gradient (src/raytracer.rs:48,67-70).  The extension follows the same trait shapes --
 * Traceable::trace(ray, min_t, max_t) -> Option<HitRecord> (src/traceable/mod.rs:36-39),
 * Material::scatter (+ an `emitted` term) -- with the semantics of the lineage the reference comes
 * from ("Ray Tracing: The Next Week": axis-aligned rects with a flip flag, a box = six rects,
 * translate / rotate-Y instances, a diffuse light).  DESIGN.md section 10 is the specification;
 * the oracle for it is this repository's own restatement (oracle/pt_oracle.c + oracle/np_oracle_ext.py):
 * no reference parity is possible for something the reference does not contain.
 * Scenes uploaded through pt_upload_scene are untouched by any of this. */
enum { PT_DIFFUSE_LIGHT = 3 };   /* pt_material.kind: scatter -> None, emitted = albedo (any value >= 0, e.g. 15) */
enum { PT_PRIM_SPHERE = 0, PT_PRIM_RECT = 1 };
/* One primitive; its id (hit-id AOV) is its index.  Sphere: v = {cx, cy, cz, radius, -}; transform
 * must be -1.  Rect: the plane `axis` = k (axis 0: a yz-rect, 1: an xz-rect, 2: an xy-rect), spanning
 * [a0, a1] x [b0, b1] on the two other axes in increasing axis order: v = {a0, a1, b0, b1, k}; its
 * normal is the unit vector of `axis`, negated when flip != 0; transform = index of a pt_transform
 * (Translate(RotateY(rect))) or -1. */
typedef struct {
    int32_t kind, material, transform, axis, flip;
    float v[5];
} pt_primitive;
/* Translate(RotateY(object)): object space -> world is a rotation about y by theta, then + offset.
 * sin / cos cross the ABI as values (the host's libm computes them, like Camera::new's tan). */
typedef struct {
    float offset[3];
    float sin_theta, cos_theta;
} pt_transform;
/* trace_ray's constants as parameters.  Defaults = the reference: max_bounces 5 (MAX_TRACES,
 * raytracer.rs:48), background_kind 0 = the sky gradient of raytracer.rs:67-68, rr_start -1 = no
 * Russian roulette.  background_kind 1 = the constant colour `background`.  rr_start >= 0: after
 * the scatter that completes bounce number b >= rr_start (b counts from 1), with q = the largest
 * attenuation component, if q < 1 one more rng.rand() is drawn; the path ends unless rand < q, else
 * attenuation /= q. */
typedef struct {
    int32_t max_bounces, background_kind;
    float background[3];
    int32_t rr_start;
} pt_integrator;
/* pt_upload_scene for a primitive list.  nodes: DFS pre-order tree over the primitives' world-space
 * boxes as for pt_upload_scene (prim = primitive index), or NULL: built here (pt_bvh_build_ext).  A
 * scene uploaded this way renders with the extension's kernel, whatever it contains. */
int pt_upload_scene_ext(pt_context *ctx, const pt_primitive *prims, int32_t n_prims,
                        const pt_material *materials, int32_t n_materials,
                        const pt_texture *textures, int32_t n_textures,
                        const pt_transform *transforms, int32_t n_transforms,
                        const pt_bvh_node *nodes, int32_t n_nodes);
/* Applies to scenes uploaded with pt_upload_scene_ext only (pt_upload_scene scenes always use the
 * reference's constants); NULL restores the defaults. */
int pt_set_integrator(pt_context *ctx, const pt_integrator *integrator);
/* pt_bvh_build over primitives: the algorithm of src/scene/bvh.rs:13-91 on world-space boxes; a rect's
 * box is padded by 0.0001 on its thin axis (zero volume would make every split cost NaN, bvh.rs:65),
 * a transformed rect's box bounds the 8 rotated corners.  Returns 2 * n_prims - 1 or a negative status. */
int32_t pt_bvh_build_ext(const pt_primitive *prims, int32_t n_prims, const pt_transform *transforms,
                         int32_t n_transforms, pt_bvh_node *out, int32_t capacity);
/* The Cornell box of "The Next Week": five walls, the light, two rotated boxes of six rects each
 * (18 primitives, 4 materials, 2 transforms).  Returns the primitive count (18), or a negative status
 * when a capacity (prims >= 18, materials >= 4, transforms >= 2) is too small.  sin / cos of the
 * two box angles (-18 and 15 degrees) are the host libm's. */
int32_t pt_gen_cornell_box(pt_primitive *prims, int32_t prim_capacity, pt_material *materials, int32_t material_capacity,
                           pt_transform *transforms, int32_t transform_capacity);

/* ---- several GPUs of one node ------------------------------------------------
 * No counterpart in the reference (one process of shared-memory threads,
 * src/parallel_renderer/mod.rs:88-145).  A group is ONE host process driving n GPUs: each holds the
 * whole scene and renders a contiguous share of the samples x samples grid -- the index
 * s = sx * samples + sy of src/raytracer.rs:77-81 is the unit that shards -- into a full-frame fp32
 * radiance buffer; one RCCL reduce (ncclReduce, sum, fp32, W*H*3 values, root = devices[0]) over
 * xGMI merges them on the render streams; the root resolves (raytracer.rs:92-93,119-121) and copies
 * the image to the host.  Per GPU the sums are the reference's sequential sums over its share; the
 * order in which the n shares are added is RCCL's: exact for n <= 2, rounding-level (<= n ulp of the
 * sum) beyond.  A group is used from one host thread at a time. */
typedef struct pt_group pt_group;
int pt_group_create(const int32_t *devices, int32_t n_devices, pt_group **out);   /* ncclCommInitAll over the devices */
void pt_group_destroy(pt_group *group);
const char *pt_group_last_error(const pt_group *group);   /* group may be NULL: create failures */
int32_t pt_group_size(const pt_group *group);
/* the context of GPU `rank` of the group, borrowed (pt_query, pt_set_lanes_per_pixel, ...); NULL if out of range */
pt_context *pt_group_context(pt_group *group, int32_t rank);
/* pt_upload_scene / pt_set_camera on every GPU of the group (the BVH is built once, on the host) */
int pt_group_upload_scene(pt_group *group, const pt_sphere *spheres, int32_t n_spheres,
                          const pt_material *materials, int32_t n_materials,
                          const pt_texture *textures, int32_t n_textures,
                          const pt_bvh_node *nodes, int32_t n_nodes);
int pt_group_set_camera(pt_group *group, const pt_camera *camera);
/* pt_upload_scene_ext / pt_set_integrator on every GPU of the group */
int pt_group_upload_scene_ext(pt_group *group, const pt_primitive *prims, int32_t n_prims,
                              const pt_material *materials, int32_t n_materials,
                              const pt_texture *textures, int32_t n_textures,
                              const pt_transform *transforms, int32_t n_transforms,
                              const pt_bvh_node *nodes, int32_t n_nodes);
int pt_group_set_integrator(pt_group *group, const pt_integrator *integrator);
/* pt_render of the full grid over the group.  Blocking.  rgb8 / radiance: host buffers as for
 * pt_render (either may be NULL); radiance is the merged sum. */
int pt_group_render(pt_group *group, int32_t samples, uint64_t seed, uint8_t *rgb8, float *radiance);
/* pt_group_render waits for the GPUs with a bounded poll (hipStreamQuery + ncclCommGetAsyncError per GPU), not
 * a blind hipStreamSynchronize: a GPU that reports an error, or has not finished within the group's timeout,
 * comes back as PT_ERR_HIP with that GPU named in pt_group_last_error; every communicator is then aborted
 * (ncclCommAbort, which releases the reduce kernels of the GPUs that did join) and the group refuses further
 * renders -- destroy it and create a new one.  Default timeout 600,000 ms; 0 = no limit.  pt_group_destroy on
 * an aborted group does not wait for a GPU whose stream never drains (that GPU's buffers are leaked instead).
 * NOTE: the n >= 2 path has been exercised by host-side fault injection and on CPU (gloo) only; no execution
 * on two or more GPUs exists yet (the build box has one). */
int pt_group_set_timeout(pt_group *group, int32_t milliseconds);
/* NCCL_VERSION_CODE the library was compiled against and what ncclGetVersion of the RCCL the process actually
 * loaded reports (under Python that is torch's bundled librccl.so, for a C++ / Rust host /opt/rocm/lib's).
 * pt_group_create refuses a loaded library of another MAJOR version (PT_ERR_HIP, both numbers in the message). */
int pt_group_rccl_version(const pt_group *group, int32_t *compiled, int32_t *loaded);
/* TEST HOOK, host side only (nothing is provoked on a GPU): the next pt_group_render's wait treats GPU `rank` as
 * reporting an asynchronous RCCL error (kind 1) or as never finishing (kind 2, until the timeout); kind 0 clears.
 * Exercises the error path of the bounded wait on a one-GPU box. */
int pt_group_inject_fault(pt_group *group, int32_t rank, int32_t kind);
/* Share of GPU `rank` of n: sample indices [*begin, *end), contiguous, sizes differ by at most one.  n < 1 or a
 * rank outside [0, n): the empty range [0, 0). */
void pt_group_shard(int32_t samples, int32_t rank, int32_t n, int32_t *begin, int32_t *end);
/* Timings of the last pt_group_render: kernel_ms[r] = GPU r's path kernel (HIP events on its stream),
 * up to `capacity` entries; *reduce_ms = from the end of the root's path kernel to the end of the
 * reduce on the root's stream (so it includes waiting for the slowest GPU).  Returns the group size. */
int pt_group_timings(const pt_group *group, float *kernel_ms, int32_t capacity, float *reduce_ms);

/* ---- host-side helpers (pure CPU, no GPU needed) ------------------------- */

/* BvhNode::construct + pre-order flatten (src/scene/bvh.rs:13-91).  Returns the
 * node count (2*n_spheres - 1) or a negative pt_status; capacity in nodes. */
int32_t pt_bvh_build(const pt_sphere *spheres, int32_t n_spheres, pt_bvh_node *out, int32_t capacity);
/* Camera::new (src/math/camera.rs:14-34), fov in degrees. */
int pt_camera_look_at(const float eye[3], const float target[3], const float up[3], float fov_degrees,
                      int32_t width, int32_t height, pt_camera *out);
/* scene::generators::simple_scene (src/scene/generators.rs:8-23): 6 spheres, 6
 * materials; material 0 expects texture 0 = decoded earthmap.  Returns 6. */
int32_t pt_gen_simple_scene(pt_sphere *spheres, pt_material *materials);
/* scene::generators::random_scene (src/scene/generators.rs:25-89) with a seeded
 * Rng::with_seed(scene_seed) in place of Rng::new() and the grid bounds
 * a in [a0,a1), b in [b0,b1) (reference: -22..22, -22..221).  One material per
 * sphere.  Returns the sphere count, or -(count needed) if capacity is too small. */
int32_t pt_gen_random_scene(uint64_t scene_seed, int32_t a0, int32_t a1, int32_t b0, int32_t b1,
                            pt_sphere *spheres, pt_material *materials, int32_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* PT_AMD_H */
