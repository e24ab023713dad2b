// pt_kernels.hip -- gfx950 kernels of the radiance hot path and their launchers.
//
//   K1 primary_hit_kernel : camera.ray + Scene::trace for one sample index (AOV)
//   K2 path_kernel        : trace_pixel + trace_ray and everything below
//                           (src/raytracer.rs:46-96), one lane per (pixel, sample chunk)
//   K3 resolve_kernel     : sqrt(sum / S^2) * 255 -> u8 (src/raytracer.rs:92-93,119-121)
//
// Compiled with -ffp-contract=off (see pt_device.hpp).
#include "pt_kernels.hpp"

using namespace ptd;

namespace {

// A 64-lane wave covers `P = 64 / lanes` pixels (a tw x th tile) times `lanes`
// sample chunks.  lane = chunk * P + pixel_slot.
__device__ __forceinline__ void tile_shape(int lanes, int &tw, int &th)
{
    switch (lanes) {
    case 1: tw = 8; th = 8; break;
    case 2: tw = 8; th = 4; break;
    case 4: tw = 4; th = 4; break;
    case 8: tw = 4; th = 2; break;
    case 16: tw = 2; th = 2; break;
    case 32: tw = 2; th = 1; break;
    default: tw = 1; th = 1; break;
    }
}

// K2.  The replacement for the tile loop of ParallelRenderer::render
// (src/parallel_renderer/mod.rs:155-203) + Raytracer::render_fragment
// (src/raytracer.rs:98-123): the grid covers the frame, each lane owns one pixel
// and a contiguous chunk of its sample indices and walks them IN ORDER, so the
// per-pixel fp32 sum is the reference's (raytracer.rs:88).
//
// The three nested loops of the reference (samples, bounces, BVH nodes) are
// flattened into ONE loop over a per-lane state machine.  Every iteration the
// wave votes (ballot + popcount, scalar) for the state most lanes are in and runs
// only that state's step, predicated on the lanes in it:
//   SPLIT  one AABB::intersect of the walk           (aabb.rs:25-36)
//   LEAF   one Sphere::trace of the walk             (sphere.rs:46-69)
//   HIT    Sphere::record + Material::scatter        (sphere.rs:37-42, material/*.rs)
//   END    sky, accumulate, next sample's camera ray (raytracer.rs:67-70,88,77-85)
// A lane whose path or walk ends never idles until the longest one of the wave
// ends, and box tests / sphere tests / shading are not serialised against each
// other inside one iteration.  Each lane still executes exactly the reference's
// operation sequence for its pixel, so results do not depend on the vote.
#ifndef PT_SPLIT_DROP
#define PT_SPLIT_DROP 2   // a SPLIT run ends once a quarter of its lanes have left the state
#endif
#ifndef PT_W_SPLIT
#define PT_W_SPLIT 1
#endif
#ifndef PT_W_LEAF
#define PT_W_LEAF 1
#endif
#ifndef PT_W_HIT
#define PT_W_HIT 1
#endif
#ifndef PT_W_END
#define PT_W_END 1
#endif

enum : int { ST_SPLIT = 0, ST_LEAF = 1, ST_HIT = 2, ST_END = 3, ST_DONE = 4 };

__global__ __launch_bounds__(64) void path_kernel(Scene sc, Camera cam, RenderParams rp, float *__restrict__ radiance)
{
    __shared__ float part[3 * 64];
    const int lane = threadIdx.x;
    const int L = rp.lanes;
    const int P = 64 / L;
    int tw, th;
    tile_shape(L, tw, th);
    const int slot = lane % P, chunk = lane / P;
    const int tiles_x = (cam.width + tw - 1) / tw;
    const int tile = blockIdx.x;
    const int px = (tile % tiles_x) * tw + slot % tw;
    const int py = (tile / tiles_x) * th + slot / tw;
    const bool valid = px < cam.width && py < cam.height;
    const uint32_t pixel = (uint32_t)(py * cam.width + px);

    const int per = (rp.s1 - rp.s0 + L - 1) / L;
    int s = rp.s0 + chunk * per;
    int s_end = s + per < rp.s1 ? s + per : rp.s1;
    if (!valid) s_end = s;

    f3 sum = fill(0.0f);
    if (L == 1 && rp.accumulate && valid)
        sum = mk(radiance[3 * pixel], radiance[3 * pixel + 1], radiance[3 * pixel + 2]);

    const int S = rp.S;
    const int n_nodes = sc.n_nodes;

    // per-lane state.  `code` is the node the walk stands on (pt_device.hpp); the
    // state of a walking lane is read off it: SPLIT, LEAF, or walk finished.
    Ray ray; ray.o = fill(0.0f); ray.d = fill(0.0f);
    f3 inv = fill(0.0f);
    float a = 0.0f;
    Hit h; h.t = T_MAX; h.prim = -1;
    int32_t code = 0;
    f3 att = fill(1.0f);
    int bounce = 0;
    uint64_t rng = 0;
    bool regular = true, to_sky = true;
    int state = ST_DONE;

    // stand on node `c`: end of walk -> HIT / END
    auto enter = [&](int32_t c) {
        code = c;
        if (code_index(c) >= n_nodes) {
            state = h.prim >= 0 ? ST_HIT : ST_END;
            to_sky = true;
        } else {
            state = c < 0 ? ST_LEAF : ST_SPLIT;
        }
    };
    auto count = [&](int st) -> int { return __builtin_amdgcn_readfirstlane((int)__popcll(__ballot(state == st))); };

    // SPLIT and LEAF steps go straight back to the vote; HIT and END fall through to
    // the shared tails: new_path -> the next sample's camera ray (or DONE),
    // new_walk -> Scene::trace prologue (scene/mod.rs:24-34) and enter(root).
    bool new_path = true, new_walk = false;
    for (;;) {
        if (new_path) {
            if (s >= s_end) state = ST_DONE;
            else {
                ray = sample_ray(cam, S, px, py, s);                             // raytracer.rs:77-85
                att = fill(1.0f);                                                // raytracer.rs:50
                bounce = 0;
                rng = path_rng(rp.seed, pixel, (uint32_t)s);
                new_walk = true;
            }
        }
        if (new_walk) {
            inv = fill(1.0f) / ray.d;                                            // scene/mod.rs:31
            regular = finite3(inv) && finite3(ray.o);
            a = dot(ray.d, ray.d);                                               // sphere.rs:49
            h.t = T_MAX; h.prim = -1;
            enter(sc.root_code);
        }
        new_path = false; new_walk = false;

        int pick;
        for (;;) {
            const int cS = count(ST_SPLIT), cL = count(ST_LEAF), cH = count(ST_HIT), cE = count(ST_END);
            int top = cS * PT_W_SPLIT, v;
            pick = ST_SPLIT;
            v = cL * PT_W_LEAF; if (v > top) { top = v; pick = ST_LEAF; }
            v = cH * PT_W_HIT;  if (v > top) { top = v; pick = ST_HIT; }
            v = cE * PT_W_END;  if (v > top) { top = v; pick = ST_END; }
            if (top == 0) pick = ST_DONE;
            if (pick == ST_SPLIT) {
                const bool all_regular = __ballot(state == ST_SPLIT && !regular) == 0ull;
                const int keep = cS - (cS >> PT_SPLIT_DROP);   // leave when this many lanes have left SPLIT
                do {
                    if (state == ST_SPLIT) {
                        const int i = code_index(code);
                        const float4 n0 = sc.nodes[2 * i], n1 = sc.nodes[2 * i + 1];
                        enter(all_regular ? split_step<true>(n0, n1, ray.o, inv, h.t)
                                          : split_step<false>(n0, n1, ray.o, inv, h.t));
                    }
                } while (count(ST_SPLIT) >= keep && keep > 0);
            } else if (pick == ST_LEAF) {
                if (state == ST_LEAF) {
                    const int i = code_index(code);
                    const float4 n0 = sc.nodes[2 * i], n1 = sc.nodes[2 * i + 1];
                    leaf_step(n0, n1, ray.o, ray.d, a, h);
                    enter(__float_as_int(n1.w));
                }
            } else break;
        }
        if (pick == ST_DONE) break;

        if (pick == ST_HIT) {
            if (state == ST_HIT) {
                // one hit iteration of `for _ in 0..MAX_TRACES` (raytracer.rs:52-61)
                f3 am; Ray nr;
                if (scatter(sc, ray, h, rng, am, nr)) {
                    att = att * am;                                              // raytracer.rs:57
                    ray = nr;
                    bounce++;
                    if (bounce >= 5) { state = ST_END; to_sky = true; }          // MAX_TRACES, raytracer.rs:48
                    else new_walk = true;
                } else {
                    state = ST_END; to_sky = false;                              // raytracer.rs:60: return 0
                }
            }
        } else {
            if (state == ST_END) {
                f3 color = fill(0.0f);
                if (to_sky) color = sky(ray, att);                               // raytracer.rs:67-70
                sum = sum + color;                                               // raytracer.rs:88
                s++;
                new_path = true;
            }
        }
    }

    if (L == 1) {
        if (valid) { radiance[3 * pixel] = sum.x; radiance[3 * pixel + 1] = sum.y; radiance[3 * pixel + 2] = sum.z; }
        return;
    }
    part[3 * lane] = sum.x; part[3 * lane + 1] = sum.y; part[3 * lane + 2] = sum.z;
    __syncthreads();
    if (chunk == 0 && valid) {
        f3 total = fill(0.0f);
        if (rp.accumulate) total = mk(radiance[3 * pixel], radiance[3 * pixel + 1], radiance[3 * pixel + 2]);
        for (int k = 0; k < L; k++) {
            const int l = k * P + slot;
            total = total + mk(part[3 * l], part[3 * l + 1], part[3 * l + 2]);
        }
        radiance[3 * pixel] = total.x; radiance[3 * pixel + 1] = total.y; radiance[3 * pixel + 2] = total.z;
    }
}

// K1: primitive id (and t) hit by the primary ray of one sample index.
__global__ __launch_bounds__(64) void primary_hit_kernel(Scene sc, Camera cam, int S, int sample, int32_t *__restrict__ hit_id,
                                                         float *__restrict__ hit_t)
{
    const int lane = threadIdx.x;
    const int tiles_x = (cam.width + 7) / 8;
    const int px = (blockIdx.x % tiles_x) * 8 + (lane & 7);
    const int py = (blockIdx.x / tiles_x) * 8 + (lane >> 3);
    if (px >= cam.width || py >= cam.height) return;
    const Ray ray = sample_ray(cam, S, px, py, sample);
    const Hit h = scene_trace(sc, ray);
    const int pixel = py * cam.width + px;
    if (hit_id) hit_id[pixel] = h.prim;
    if (hit_t) hit_t[pixel] = h.t;
}

// K3: raytracer.rs:92-93 (sum / S^2, sqrt) and :119-121 (*255, `as u8`).
__global__ __launch_bounds__(256) void resolve_kernel(const float *__restrict__ radiance, int n_pixels, float n_samples,
                                                      uint8_t *__restrict__ rgb8)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    f3 c = mk(radiance[3 * i], radiance[3 * i + 1], radiance[3 * i + 2]) / n_samples;
    c = mk(__builtin_sqrtf(c.x), __builtin_sqrtf(c.y), __builtin_sqrtf(c.z)) * 255.0f;
    rgb8[3 * i] = trunc_sat_u8(c.x);
    rgb8[3 * i + 1] = trunc_sat_u8(c.y);
    rgb8[3 * i + 2] = trunc_sat_u8(c.z);
}

}  // namespace

namespace ptk {

int tiles_for(const Camera &cam, int lanes)
{
    int tw = 8, th = 8;
    switch (lanes) {
    case 1: tw = 8; th = 8; break;
    case 2: tw = 8; th = 4; break;
    case 4: tw = 4; th = 4; break;
    case 8: tw = 4; th = 2; break;
    case 16: tw = 2; th = 2; break;
    case 32: tw = 2; th = 1; break;
    default: tw = 1; th = 1; break;
    }
    return ((cam.width + tw - 1) / tw) * ((cam.height + th - 1) / th);
}

hipError_t launch_path(const Scene &sc, const Camera &cam, const RenderParams &rp, float *d_radiance, hipStream_t stream)
{
    const int tiles = tiles_for(cam, rp.lanes);
    hipLaunchKernelGGL(path_kernel, dim3(tiles), dim3(64), 0, stream, sc, cam, rp, d_radiance);
    return hipGetLastError();
}

hipError_t launch_primary_hits(const Scene &sc, const Camera &cam, int S, int sample, int32_t *d_hit_id, float *d_hit_t,
                               hipStream_t stream)
{
    const int tiles = tiles_for(cam, 1);
    hipLaunchKernelGGL(primary_hit_kernel, dim3(tiles), dim3(64), 0, stream, sc, cam, S, sample, d_hit_id, d_hit_t);
    return hipGetLastError();
}

hipError_t launch_resolve(const float *d_radiance, int n_pixels, int S, uint8_t *d_rgb8, hipStream_t stream)
{
    const int blocks = (n_pixels + 255) / 256;
    hipLaunchKernelGGL(resolve_kernel, dim3(blocks), dim3(256), 0, stream, d_radiance, n_pixels, (float)(S * S), d_rgb8);
    return hipGetLastError();
}

}  // namespace ptk
