// pt_group.cpp -- the multi-GPU half of the C ABI (include/pt_amd.h, "group" section).
//
// One host process drives several GPUs of a node: every GPU holds the whole scene (1-3 MB) and
// renders a contiguous share of the samples x samples grid -- the index s = sx * S + sy of
// src/raytracer.rs:77-81 is the unit that shards -- into a full-frame fp32 radiance buffer; one
// RCCL reduce (sum, fp32, W*H*3 values, root = the group's first GPU) over xGMI merges the buffers
// on their render streams; the root resolves (raytracer.rs:92-93,119-121) and copies the image out.
// The reference has no counterpart: it is one process of shared-memory threads
// (src/parallel_renderer/mod.rs:88-145).
//
// Only the root needs the sum, so this is a reduce, not an all-reduce: with 7 direct xGMI links per
// GPU every peer's 24.9 MB (1080p) rides its own link to the root, a fraction of a millisecond next
// to a frame's hundreds.  No exception leaves this file.
#include <rccl/rccl.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <new>
#include <thread>

#include "pt_context.hpp"

struct pt_group {
    std::vector<pt_context *> ctx;      // one context (device binding, stream, scene copy) per GPU
    std::vector<ncclComm_t> comm;       // ncclCommInitAll: one communicator per GPU, same order
    std::vector<float> kernel_ms;       // per GPU: path kernel of the last pt_group_render
    hipEvent_t ev_reduce0 = nullptr, ev_reduce1 = nullptr;   // on the root's stream, around the reduce
    float reduce_ms = -1.0f;
    std::string error;
    int32_t timeout_ms = 600000;        // pt_group_set_timeout: longest wait for a frame (0 = no limit)
    bool aborted = false;               // a render failed or timed out: the communicators are gone (ncclCommAbort)
    int rccl_compiled = NCCL_VERSION_CODE, rccl_loaded = 0;
    // test hook (pt_group_inject_fault): the next render's wait treats `inject_rank` as failed (1) or as never finishing (2)
    int inject_rank = -1, inject_kind = 0;
};

namespace {

thread_local std::string g_group_create_error;

int gfail(pt_group *g, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
int gfail(pt_group *g, int code, const char *fmt, ...)
{
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->error = buf;
    else g_group_create_error = buf;
    return code;
}

// a rank's failure, with the rank named
int rank_fail(pt_group *g, int rank, int code) { return gfail(g, code, "GPU %d of the group (device %d): %s", rank, g->ctx[rank]->device, pt_last_error(g->ctx[rank])); }

// After a failed or timed-out frame: every communicator is aborted (a collective that a peer never joined would
// otherwise keep its kernel spinning on the other GPUs), and the group refuses further renders.
void abort_group(pt_group *g)
{
    for (size_t r = 0; r < g->comm.size(); r++)
        if (g->comm[r]) { (void)ncclCommAbort(g->comm[r]); g->comm[r] = nullptr; }
    g->aborted = true;
}

// Has everything enqueued on the context's render stream finished?  Waits at most `ms` milliseconds.
bool stream_idle_within(pt_context *c, int ms)
{
    if (hipSetDevice(c->device) != hipSuccess) return false;
    const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(ms);
    for (;;) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return true;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); return false; }
        if (std::chrono::steady_clock::now() >= until) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

void destroy(pt_group *g)
{
    for (size_t r = 0; r < g->comm.size(); r++)
        if (g->comm[r]) (void)ncclCommDestroy(g->comm[r]);
    if (!g->ctx.empty() && g->ctx[0]) (void)hipSetDevice(g->ctx[0]->device);
    if (g->ev_reduce0) (void)hipEventDestroy(g->ev_reduce0);
    if (g->ev_reduce1) (void)hipEventDestroy(g->ev_reduce1);
    for (pt_context *c : g->ctx) {
        if (!c) continue;
        // pt_destroy waits for the context's stream and frees its buffers (hipFree waits for the device): after an
        // abort a GPU whose work never finishes would hold the caller there for ever -- such a context is leaked
        // (its GPU is lost to this process anyway) rather than waited for
        if (g->aborted && !stream_idle_within(c, 2000)) continue;
        pt_destroy(c);
    }
    delete g;
}

}  // namespace

extern "C" {

void pt_group_shard(int32_t samples, int32_t rank, int32_t n, int32_t *begin, int32_t *end)
{
    if (n < 1 || rank < 0 || rank >= n || samples < 0) {   // no such share: the empty range
        if (begin) *begin = 0;
        if (end) *end = 0;
        return;
    }
    const int64_t total = (int64_t)samples * samples;
    const int64_t base = total / n, rem = total % n;
    const int64_t b = rank * base + (rank < rem ? rank : rem);
    if (begin) *begin = (int32_t)b;
    if (end) *end = (int32_t)(b + base + (rank < rem ? 1 : 0));
}

int pt_group_create(const int32_t *devices, int32_t n_devices, pt_group **out)
{
    if (!out) return gfail(nullptr, PT_ERR_INVALID, "pt_group_create: out is NULL");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return gfail(nullptr, PT_ERR_INVALID, "pt_group_create: 1..64 devices, got %d", n_devices);
    for (int32_t i = 0; i < n_devices; i++)
        for (int32_t j = 0; j < i; j++)
            if (devices[i] == devices[j]) return gfail(nullptr, PT_ERR_INVALID, "pt_group_create: device %d listed twice", devices[i]);
    pt_group *g = new (std::nothrow) pt_group();
    if (!g) return gfail(nullptr, PT_ERR_INVALID, "out of host memory");
    try {
        g->ctx.assign((size_t)n_devices, nullptr);
        g->comm.assign((size_t)n_devices, nullptr);
        g->kernel_ms.assign((size_t)n_devices, -1.0f);
    } catch (...) {
        delete g;
        return gfail(nullptr, PT_ERR_INVALID, "out of host memory");
    }
    for (int32_t r = 0; r < n_devices; r++) {
        const int rc = pt_create(devices[r], &g->ctx[r]);
        if (rc != PT_OK) {
            gfail(nullptr, rc, "device %d: %s", devices[r], pt_last_error(nullptr));
            destroy(g);
            return rc;
        }
    }
    // Which RCCL answers is decided by what the process loaded first (under Python: torch's bundled librccl.so;
    // a C++ / Rust host: /opt/rocm/lib's): the library that runs must be of the major version this file was
    // compiled against, or its ncclComm_t / ncclResult_t may not mean what the header says.
    int loaded = 0;
    if (ncclGetVersion(&loaded) != ncclSuccess) loaded = 0;
    g->rccl_loaded = loaded;
    const int major_of_loaded = loaded >= 10000 ? loaded / 10000 : loaded / 1000;
    if (major_of_loaded != NCCL_MAJOR) {
        gfail(nullptr, PT_ERR_HIP, "RCCL version mismatch: compiled against %d (NCCL_VERSION_CODE), the loaded library reports %d",
              (int)NCCL_VERSION_CODE, loaded);
        destroy(g);
        return PT_ERR_HIP;
    }
    // one communicator per GPU in one process (RCCL over xGMI)
    std::vector<int> devlist(devices, devices + n_devices);
    const ncclResult_t nr = ncclCommInitAll(g->comm.data(), n_devices, devlist.data());
    if (nr != ncclSuccess) {
        gfail(nullptr, PT_ERR_HIP, "ncclCommInitAll over %d GPUs failed: %s", n_devices, ncclGetErrorString(nr));
        for (auto &c : g->comm) c = nullptr;
        destroy(g);
        return PT_ERR_HIP;
    }
    hipError_t e = hipSetDevice(devices[0]);
    if (e == hipSuccess) e = hipEventCreate(&g->ev_reduce0);
    if (e == hipSuccess) e = hipEventCreate(&g->ev_reduce1);
    if (e != hipSuccess) {
        gfail(nullptr, PT_ERR_HIP, "event creation on device %d failed: %s", devices[0], hipGetErrorString(e));
        destroy(g);
        return PT_ERR_HIP;
    }
    *out = g;
    return PT_OK;
}

void pt_group_destroy(pt_group *g)
{
    if (!g) return;
    if (!g->aborted)
        for (pt_context *c : g->ctx)
            if (c) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
    destroy(g);
}

int pt_group_set_timeout(pt_group *g, int32_t milliseconds)
{
    if (!g) return PT_ERR_INVALID;
    if (milliseconds < 0) return gfail(g, PT_ERR_INVALID, "timeout must be >= 0 milliseconds (0 = no limit), got %d", milliseconds);
    g->timeout_ms = milliseconds;
    return PT_OK;
}

int pt_group_rccl_version(const pt_group *g, int32_t *compiled, int32_t *loaded)
{
    if (!g) return PT_ERR_INVALID;
    if (compiled) *compiled = g->rccl_compiled;
    if (loaded) *loaded = g->rccl_loaded;
    return PT_OK;
}

int pt_group_inject_fault(pt_group *g, int32_t rank, int32_t kind)
{
    if (!g) return PT_ERR_INVALID;
    if (kind != 0 && (rank < 0 || rank >= (int32_t)g->ctx.size() || kind < 1 || kind > 2))
        return gfail(g, PT_ERR_INVALID, "pt_group_inject_fault: rank in [0, %d) and kind 0 (none), 1 (error) or 2 (never finishes)", (int)g->ctx.size());
    g->inject_rank = kind ? rank : -1;
    g->inject_kind = kind;
    return PT_OK;
}

const char *pt_group_last_error(const pt_group *g) { return g ? g->error.c_str() : g_group_create_error.c_str(); }

int32_t pt_group_size(const pt_group *g) { return g ? (int32_t)g->ctx.size() : 0; }

pt_context *pt_group_context(pt_group *g, int32_t rank)
{
    if (!g || rank < 0 || rank >= (int32_t)g->ctx.size()) return nullptr;
    return g->ctx[rank];
}

int pt_group_upload_scene(pt_group *g, const pt_sphere *spheres, int32_t n_spheres, const pt_material *materials,
                          int32_t n_materials, const pt_texture *textures, int32_t n_textures, const pt_bvh_node *nodes,
                          int32_t n_nodes)
{
    if (!g) return PT_ERR_INVALID;
    // build the tree once on the host (bvh.rs:13-91), then hand every GPU the same flattened nodes
    std::vector<pt_bvh_node> built;
    if ((!nodes || n_nodes == 0) && spheres && n_spheres > 0) {
        try {
            built.resize(2 * (size_t)n_spheres - 1);
        } catch (...) {
            return gfail(g, PT_ERR_INVALID, "out of host memory");
        }
        const int32_t got = pt_bvh_build(spheres, n_spheres, built.data(), (int32_t)built.size());
        if (got != (int32_t)built.size()) return gfail(g, got < 0 ? got : PT_ERR_INVALID, "BVH build failed (%d)", got);
        nodes = built.data();
        n_nodes = got;
    }
    for (size_t r = 0; r < g->ctx.size(); r++) {
        const int rc = pt_upload_scene(g->ctx[r], spheres, n_spheres, materials, n_materials, textures, n_textures, nodes, n_nodes);
        if (rc != PT_OK) return rank_fail(g, (int)r, rc);
    }
    return PT_OK;
}

int pt_group_upload_scene_ext(pt_group *g, const pt_primitive *prims, int32_t n_prims, const pt_material *materials,
                              int32_t n_materials, const pt_texture *textures, int32_t n_textures,
                              const pt_transform *transforms, int32_t n_transforms, const pt_bvh_node *nodes, int32_t n_nodes)
{
    if (!g) return PT_ERR_INVALID;
    std::vector<pt_bvh_node> built;
    if ((!nodes || n_nodes == 0) && prims && n_prims > 0) {
        try {
            built.resize(2 * (size_t)n_prims - 1);
        } catch (...) {
            return gfail(g, PT_ERR_INVALID, "out of host memory");
        }
        const int32_t got = pt_bvh_build_ext(prims, n_prims, transforms, n_transforms, built.data(), (int32_t)built.size());
        if (got != (int32_t)built.size()) return gfail(g, got < 0 ? got : PT_ERR_INVALID, "BVH build over the primitives failed (%d)", got);
        nodes = built.data();
        n_nodes = got;
    }
    for (size_t r = 0; r < g->ctx.size(); r++) {
        const int rc = pt_upload_scene_ext(g->ctx[r], prims, n_prims, materials, n_materials, textures, n_textures, transforms,
                                           n_transforms, nodes, n_nodes);
        if (rc != PT_OK) return rank_fail(g, (int)r, rc);
    }
    return PT_OK;
}

int pt_group_set_integrator(pt_group *g, const pt_integrator *integrator)
{
    if (!g) return PT_ERR_INVALID;
    for (size_t r = 0; r < g->ctx.size(); r++) {
        const int rc = pt_set_integrator(g->ctx[r], integrator);
        if (rc != PT_OK) return rank_fail(g, (int)r, rc);
    }
    return PT_OK;
}

int pt_group_set_camera(pt_group *g, const pt_camera *camera)
{
    if (!g) return PT_ERR_INVALID;
    for (size_t r = 0; r < g->ctx.size(); r++) {
        const int rc = pt_set_camera(g->ctx[r], camera);
        if (rc != PT_OK) return rank_fail(g, (int)r, rc);
    }
    return PT_OK;
}

int pt_group_render(pt_group *g, int32_t samples, uint64_t seed, uint8_t *rgb8, float *radiance)
{
    if (!g) return PT_ERR_INVALID;
    const int n = (int)g->ctx.size();
    if (samples < 2 || samples > 1024) return gfail(g, PT_ERR_INVALID, "samples must be in [2, 1024] (got %d)", samples);
    if ((int64_t)samples * samples < n) return gfail(g, PT_ERR_INVALID, "%d samples cannot be shared by %d GPUs", samples * samples, n);
    if (g->aborted) return gfail(g, PT_ERR_HIP, "this group was aborted by an earlier failure (%s): destroy it and create a new one", g->error.c_str());
    pt_context *root = g->ctx[0];
    // every GPU must render the root's frame: the reduce reads W*H*3 values of every rank's buffer (a rank whose
    // camera was changed through pt_group_context would hand RCCL a shorter one)
    for (int r = 0; r < n; r++) {
        const pt_context *c = g->ctx[r];
        if (!c->have_camera || !root->have_camera) return gfail(g, PT_ERR_NO_SCENE, "GPU %d of the group has no camera (pt_group_set_camera)", r);
        if (c->camera.width != root->camera.width || c->camera.height != root->camera.height)
            return gfail(g, PT_ERR_INVALID, "GPU %d of the group renders %d x %d, the root %d x %d: set cameras through pt_group_set_camera",
                         r, c->camera.width, c->camera.height, root->camera.width, root->camera.height);
    }
    // 1. every GPU renders its share of the sample grid into its own full-frame buffer (asynchronous)
    for (int r = 0; r < n; r++) {
        pt_context *c = g->ctx[r];
        if (hipSetDevice(c->device) != hipSuccess) return gfail(g, PT_ERR_HIP, "hipSetDevice(%d) failed", c->device);
        int rc = ptapi::ensure_frame(c);
        if (rc != PT_OK) return rank_fail(g, r, rc);
        if (c->frame_pixels != root->frame_pixels) return gfail(g, PT_ERR_INVALID, "GPU %d of the group holds a frame of another size than the root's", r);
        int32_t s0, s1;
        pt_group_shard(samples, r, n, &s0, &s1);
        rc = pt_render_device(c, samples, seed, s0, s1, 0, c->d_radiance, nullptr);
        if (rc != PT_OK) return rank_fail(g, r, rc);
    }
    // 2. one reduce to the root, enqueued behind each GPU's path kernel on its render stream
    const size_t count = (size_t)root->frame_pixels * 3;
    if (hipSetDevice(root->device) != hipSuccess || hipEventRecord(g->ev_reduce0, root->stream) != hipSuccess)
        return gfail(g, PT_ERR_HIP, "event record on the root failed");
    ncclResult_t nr = ncclGroupStart();
    for (int r = 0; r < n && nr == ncclSuccess; r++) {
        pt_context *c = g->ctx[r];
        nr = ncclReduce(c->d_radiance, c->d_radiance, count, ncclFloat32, ncclSum, 0, g->comm[r], c->stream);
    }
    const ncclResult_t ne = ncclGroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) {
        gfail(g, PT_ERR_HIP, "ncclReduce of %zu floats over %d GPUs failed: %s", count, n, ncclGetErrorString(nr));
        abort_group(g);
        return PT_ERR_HIP;
    }
    if (hipSetDevice(root->device) != hipSuccess || hipEventRecord(g->ev_reduce1, root->stream) != hipSuccess)
        return gfail(g, PT_ERR_HIP, "event record on the root failed");
    // 3. the root resolves and hands the image over
    if (rgb8) {
        const int rc = pt_resolve_device(root, samples, root->d_radiance, root->d_rgb8, nullptr);
        if (rc != PT_OK) return rank_fail(g, 0, rc);
        if (hipMemcpyAsync(rgb8, root->d_rgb8, count, hipMemcpyDeviceToHost, root->stream) != hipSuccess)
            return gfail(g, PT_ERR_HIP, "copy of the image to the host failed");
    }
    if (radiance && hipMemcpyAsync(radiance, root->d_radiance, count * sizeof(float), hipMemcpyDeviceToHost, root->stream) != hipSuccess)
        return gfail(g, PT_ERR_HIP, "copy of the radiance to the host failed");
    // 4. a BOUNDED wait for every GPU's stream.  A peer that faults, or never joins the reduce, must come back
    // as a status code that names it, not as a caller blocked in hipStreamSynchronize for ever: every stream is
    // polled (hipStreamQuery) together with its communicator's asynchronous error state; on an error or when the
    // group's timeout runs out all communicators are aborted (which releases the reduce kernels of the GPUs that
    // did join) and the group refuses further renders.
    {
        std::vector<char> finished;
        try {
            finished.assign((size_t)n, 0);
        } catch (...) {
            abort_group(g);
            return gfail(g, PT_ERR_INVALID, "out of host memory");
        }
        const auto t0 = std::chrono::steady_clock::now();
        const int inject_rank = g->inject_rank, inject_kind = g->inject_kind;
        g->inject_rank = -1; g->inject_kind = 0;            // a test hook fires once
        int left = n, spins = 0;
        while (left > 0) {
            for (int r = 0; r < n; r++) {
                if (finished[(size_t)r]) continue;
                pt_context *c = g->ctx[r];
                ncclResult_t async = ncclSuccess;
                ncclResult_t ar = ncclCommGetAsyncError(g->comm[r], &async);
                if (r == inject_rank && inject_kind == 1) { ar = ncclSuccess; async = ncclSystemError; }
                if (ar != ncclSuccess || (async != ncclSuccess && async != ncclInProgress)) {
                    gfail(g, PT_ERR_HIP, "GPU %d of the group (device %d): RCCL reports an asynchronous error (%s); the group was aborted",
                          r, c->device, ncclGetErrorString(ar != ncclSuccess ? ar : async));
                    abort_group(g);
                    return PT_ERR_HIP;
                }
                hipError_t q = hipSetDevice(c->device);
                if (q == hipSuccess) q = hipStreamQuery(c->stream);
                if (r == inject_rank && inject_kind == 2 && q == hipSuccess) q = hipErrorNotReady;
                if (q == hipSuccess) { finished[(size_t)r] = 1; left--; continue; }
                if (q != hipErrorNotReady) {
                    gfail(g, PT_ERR_HIP, "GPU %d of the group (device %d): %s; the group was aborted", r, c->device, hipGetErrorString(q));
                    (void)hipGetLastError();
                    abort_group(g);
                    return PT_ERR_HIP;
                }
            }
            if (left == 0) break;
            if (g->timeout_ms > 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(g->timeout_ms)) {
                int late = 0;
                while (late < n && finished[(size_t)late]) late++;
                gfail(g, PT_ERR_HIP, "GPU %d of the group (device %d) had not finished its frame after %d ms (%d of %d GPUs had); the group was aborted",
                      late, g->ctx[late]->device, g->timeout_ms, n - left, n);
                abort_group(g);
                return PT_ERR_HIP;
            }
            // a frame takes milliseconds to seconds: yield for the first polls, then sleep 50 us between them
            if (++spins < 64) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    for (int r = 0; r < n; r++) {
        g->kernel_ms[r] = pt_last_path_kernel_ms(g->ctx[r]);
        if (g->kernel_ms[r] == -2.0f) return rank_fail(g, r, PT_ERR_HIP);
        if (g->kernel_ms[r] < 0.0f) {
            // the events could not be read, so the kernel's error word was not looked at either: read it now
            const int rc = ptapi::check_kernel_error(g->ctx[r]);
            if (rc != PT_OK) return rank_fail(g, r, rc);
            return gfail(g, PT_ERR_HIP, "GPU %d of the group (device %d): the path kernel's events could not be read", r, g->ctx[r]->device);
        }
    }
    g->reduce_ms = -1.0f;
    (void)hipSetDevice(root->device);
    (void)hipEventElapsedTime(&g->reduce_ms, g->ev_reduce0, g->ev_reduce1);
    return PT_OK;
}

int pt_group_timings(const pt_group *g, float *kernel_ms, int32_t capacity, float *reduce_ms)
{
    if (!g) return PT_ERR_INVALID;
    const int32_t n = (int32_t)g->ctx.size();
    if (kernel_ms)
        for (int32_t r = 0; r < n && r < capacity; r++) kernel_ms[r] = g->kernel_ms[r];
    if (reduce_ms) *reduce_ms = g->reduce_ms;
    return n;
}

}  // extern "C"
