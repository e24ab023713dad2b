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
// and a contiguous chunk of its sample indices and walks them in order.  The
// sample loop and the bounce loop of trace_pixel / trace_ray are flattened into
// ONE loop whose body is one path segment: a lane whose path ends starts its
// next sample at once instead of idling until the longest path of the wave ends.
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
    Ray ray; ray.o = fill(0.0f); ray.d = fill(0.0f);
    f3 att = fill(1.0f);
    int bounce = 0;
    uint64_t rng = 0;
    bool alive = false;

    for (;;) {
        if (!alive) {
            if (s >= s_end) break;
            ray = sample_ray(cam, S, px, py, s);                                 // raytracer.rs:77-85
            att = fill(1.0f);                                                    // raytracer.rs:50
            bounce = 0;
            rng = path_rng(rp.seed, pixel, (uint32_t)s);
            alive = true;
        }
        // one iteration of the `for _ in 0..MAX_TRACES` loop (raytracer.rs:52-65)
        const Hit h = scene_trace(sc, ray);
        bool to_sky = true;
        bool done = true;
        f3 color = fill(0.0f);
        if (h.prim >= 0) {
            f3 a; Ray nr;
            if (scatter(sc, ray, h, rng, a, nr)) {
                att = att * a;                                                   // raytracer.rs:57
                ray = nr;
                bounce++;
                done = bounce >= 5;                                              // MAX_TRACES, raytracer.rs:48
            } else {
                to_sky = false;                                                  // raytracer.rs:60: return 0
            }
        }
        if (done) {
            if (to_sky) color = sky(ray, att);                                   // raytracer.rs:67-70
            sum = sum + color;                                                   // raytracer.rs:88
            s++;
            alive = false;
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
