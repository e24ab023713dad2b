// pt_context.hpp -- the context behind include/pt_amd.h's opaque pt_context, shared by the
// translation units that implement the C ABI (pt_api.cpp: one GPU; pt_group.cpp: several).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/pt_amd.h"
#include "pt_kernels.hpp"

struct pt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    std::string error;

    // device scene
    float4 *d_nodes = nullptr;
    float4 *d_prims = nullptr;
    int32_t *d_prim_ids = nullptr;   // caller's primitive id per device index (= rank in the reference's walk order)
    float4 *d_xforms = nullptr;      // extension scenes: two float4 per pt_transform
    ptd::Texture *d_textures = nullptr;
    std::vector<uint8_t *> d_texels;
    ptd::Scene scene{};
    bool have_scene = false;

    pt_integrator integrator{5, 0, {0.0f, 0.0f, 0.0f}, -1};   // pt_set_integrator; read by extension scenes only

    ptd::Camera camera{};
    bool have_camera = false;

    // frame-sized scratch for the host-buffer entry point pt_render
    float *d_radiance = nullptr;
    uint8_t *d_rgb8 = nullptr;
    int32_t *d_hit_id = nullptr;
    int64_t frame_pixels = 0;

    int32_t lanes_override = 0;
    int32_t mode = PT_EXACT;      // pt_set_mode

    // the uploaded tree (DFS pre-order, validated) and its leaves' device records (two float4 per node), kept so
    // that the node arrays can be threaded again when the camera or the walk order changes (pt_api.cpp thread_nodes)
    std::vector<pt_bvh_node> h_tree;
    std::vector<float4> h_leaf;
    std::vector<float4> h_threaded;      // host copy of what d_nodes holds (n0[] then n1[]): an unchanged threading is not uploaded again
    int32_t walk_order = PT_WALK_VIEW;   // pt_set_walk_order
    bool view_ok = false;                // not an extension scene
    bool threaded = false, threaded_view = false;   // what d_nodes holds now
    float threaded_eye[3] = {0.0f, 0.0f, 0.0f};

    // primary-ray candidate lists (pt_api.cpp build_primary_lists; Scene::cand)
    int32_t *d_leaf_codes = nullptr;     // per primitive rank: the code of its Leaf in the current threading
    int32_t *d_lists = nullptr;          // per 8x8 pixel block of the camera's frame: PT_PRIMARY_K codes
    size_t lists_bytes = 0;
    bool lists_valid = false;            // d_lists was built for lists_camera and the current node codes
    ptd::Camera lists_camera{};
    bool primary_lists = true;           // pt_set_primary_lists

    int32_t *d_sched = nullptr;   // work counter, error flag, per-tile progress (pt_kernels.hpp SchedParams)
    size_t sched_bytes = 0;
    // the device's copy of the library's file-scope scratch array (pt_render_device_static), resolved in pt_create
    int32_t *static_sched = nullptr;
    size_t static_sched_bytes = 0;
    int32_t *last_sched = nullptr;   // the scratch the most recent path-kernel launch used (its error word)
    hipStream_t last_stream = nullptr;   // and the stream it went to
    // in-flight record of the context's OWN scratch (d_sched): an event behind its most recent launch and that
    // launch's stream.  (The library's static array has its record in pt_kernels.hip, shared by the GPU's contexts.)
    hipEvent_t ev_own_done = nullptr;
    hipStream_t own_stream = nullptr;
    bool own_used = false;
};

namespace ptapi {
// records the message on the context (or for pt_last_error(NULL) when ctx is NULL) and returns `code`
int fail(pt_context *ctx, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
// frame-sized device buffers of the context (radiance, rgb8, hit ids) for its current camera
int ensure_frame(pt_context *c);
// reads the error word the most recent path-kernel launch left (the launch must have completed)
int check_kernel_error(pt_context *c);
}  // namespace ptapi

#define PT_HIP(ctx, call)                                                                                          \
    do {                                                                                                           \
        hipError_t e_ = (call);                                                                                    \
        if (e_ != hipSuccess) return ptapi::fail((ctx), PT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)
