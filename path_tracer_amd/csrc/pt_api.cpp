// pt_api.cpp -- the extern "C" boundary of libpt_amd.so (include/pt_amd.h).
// Context / upload / launch plumbing only; the arithmetic lives in pt_device.hpp
// and the kernels in pt_kernels.hip.  No exception leaves this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "pt_context.hpp"

namespace {

thread_local std::string g_create_error;

// lanes needed to give every SIMD of the chip a full complement of waves:
// 256 CU x 4 SIMD x 8 waves x 64 lanes (MI355X_MICROARCH.md "Chip-level parameters")
constexpr int64_t kFullChipLanes = 256ll * 4 * 8 * 64;
// n0[] entries (16 B each) a workgroup stages in LDS: 16 KiB, so LDS (160 KiB per CU)
// never limits residency below the 8 workgroups x 4 waves a CU can hold
constexpr int32_t kMaxLdsNodes = PT_MAX_LDS_NODES;   // both halves of these nodes sit in LDS (pt_kernels.hpp: 64 KiB per group, the waves' chains at the top)
// Slots of the device node arrays: a tree that fits is numbered 0 .. n-1; a larger one keeps kMaxLdsNodes staged nodes in
// slots 0 .. kMaxLdsNodes-1 and the others from PT_FIRST_GLOBAL_SLOT on (the slots between are never used), so that a
// byte offset below PT_LDS_N1_OFFSET always means LDS.
int32_t slots_for(int32_t n_nodes) { return n_nodes <= kMaxLdsNodes ? n_nodes : PT_FIRST_GLOBAL_SLOT + (n_nodes - kMaxLdsNodes); }
// 16 * end_slot_for(n) is the code that ends a walk: above every node's code and above the chain entries' LDS offsets
int32_t end_slot_for(int32_t n_nodes) { return std::max(slots_for(n_nodes), (int32_t)PT_FIRST_GLOBAL_SLOT); }

}  // namespace

namespace ptapi {

int fail(pt_context *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf;
    else g_create_error = buf;
    return code;
}

}  // namespace ptapi

using ptapi::fail;

namespace {
void free_frame(pt_context *c);
}

namespace ptapi {

int ensure_frame(pt_context *c)
{
    const int64_t px = (int64_t)c->camera.width * c->camera.height;
    if (px == c->frame_pixels) return PT_OK;
    free_frame(c);
    PT_HIP(c, hipMalloc((void **)&c->d_radiance, (size_t)px * 3 * sizeof(float)));
    PT_HIP(c, hipMalloc((void **)&c->d_rgb8, (size_t)px * 3));
    PT_HIP(c, hipMalloc((void **)&c->d_hit_id, (size_t)px * sizeof(int32_t)));
    c->frame_pixels = px;
    return PT_OK;
}

// the kernel's bounded dependency wait (pt_kernels.hip) reports here instead of hanging
int check_kernel_error(pt_context *c)
{
    if (!c->last_sched) return PT_OK;
#ifdef PT_PHASES
    {   // measurement build: cycle sums the persistent kernel left in the control words
        unsigned long long w[16] = {0};
        if (hipMemcpy(w, c->last_sched, sizeof w, hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "PT_PHASES items %d waves %llu draw %llu item %llu wait %llu\n", (int)(w[0] & 0xffffffffu), w[10], w[8], w[9], w[11]);
    }
#endif
    int32_t ctrl[2] = {0, 0};
    PT_HIP(c, hipMemcpy(ctrl, c->last_sched, sizeof ctrl, hipMemcpyDeviceToHost));
    if (ctrl[1] != 0) {
        // a launch that reported an error may have left its words in any state: the whole scratch is cleared
        // (the launch is complete, and its stream idle, when this runs)
        const size_t bytes = c->last_sched == c->static_sched ? c->static_sched_bytes : c->sched_bytes;
        (void)hipMemset(c->last_sched, 0, bytes);
        if (ctrl[1] == 2) return fail(c, PT_ERR_HIP, "path kernel: dynamic LDS does not start at address 0");
        return fail(c, PT_ERR_HIP, "path kernel: a piece gave up waiting for its predecessor");
    }
    return PT_OK;
}

}  // namespace ptapi

using ptapi::ensure_frame;

namespace {

void free_scene(pt_context *c)
{
    if (c->d_nodes) (void)hipFree(c->d_nodes);
    if (c->d_prims) (void)hipFree(c->d_prims);
    if (c->d_prim_ids) (void)hipFree(c->d_prim_ids);
    c->d_prim_ids = nullptr;
    if (c->d_leaf_codes) (void)hipFree(c->d_leaf_codes);
    c->d_leaf_codes = nullptr;
    if (c->d_textures) (void)hipFree(c->d_textures);
    if (c->d_xforms) (void)hipFree(c->d_xforms);
    for (uint8_t *p : c->d_texels) (void)hipFree(p);
    c->d_nodes = c->d_prims = c->d_xforms = nullptr;
    c->d_textures = nullptr;
    c->d_texels.clear();
    c->have_scene = false;
}

void free_frame(pt_context *c)
{
    if (c->d_radiance) (void)hipFree(c->d_radiance);
    if (c->d_rgb8) (void)hipFree(c->d_rgb8);
    if (c->d_hit_id) (void)hipFree(c->d_hit_id);
    c->d_radiance = nullptr;
    c->d_rgb8 = nullptr;
    c->d_hit_id = nullptr;
    c->frame_pixels = 0;
}

int check_ready(pt_context *c, int32_t samples)
{
    if (!c) return PT_ERR_INVALID;
    if (!c->have_scene) return fail(c, PT_ERR_NO_SCENE, "no scene uploaded (pt_upload_scene)");
    if (!c->have_camera) return fail(c, PT_ERR_NO_SCENE, "no camera set (pt_set_camera)");
    // samples == 1 divides 0/0 in the reference's grid (raytracer.rs:78): refuse it
    if (samples < 2) return fail(c, PT_ERR_INVALID, "samples must be >= 2 (got %d)", samples);
    if (samples > 1024) return fail(c, PT_ERR_LIMIT, "samples must be <= 1024 (got %d)", samples);
    return PT_OK;
}

// Does every Split's box contain the boxes of its two children, and every sphere's box (centre -+ |radius|) its
// sphere?  "The order of the walk does not change a ray's closest hit" rests on it: a box that does not
// contain what is below it can hide a closer hit behind a farther one found earlier, and then the order matters.
// The reference builds such boxes for spheres of NEGATIVE radius (Sphere::bounding_box is centre -+ radius as
// written, so min and max swap: the hollow-glass idiom); a caller's tree may hold anything.
bool boxes_contain(const pt_bvh_node *nodes, int32_t n_nodes, const pt_primitive *prims)
{
    auto inside = [](const float *lo, const float *hi, const float *plo, const float *phi) {
        for (int k = 0; k < 3; k++)
            if (!(plo[k] <= lo[k] && hi[k] <= phi[k])) return false;   // NaN fails
        return true;
    };
    for (int32_t i = 0; i < n_nodes; i++) {
        if (nodes[i].prim >= 0) continue;
        const int32_t kids[2] = {i + 1, nodes[i + 1].skip};
        for (int32_t kid : kids) {
            float lo[3], hi[3];
            if (nodes[kid].prim >= 0) {
                const pt_primitive &p = prims[nodes[kid].prim];
                if (p.kind != PT_PRIM_SPHERE) return false;
                const float r = std::fabs(p.v[3]);
                for (int k = 0; k < 3; k++) { lo[k] = p.v[k] - r; hi[k] = p.v[k] + r; }
            } else {
                for (int k = 0; k < 3; k++) { lo[k] = nodes[kid].bmin[k]; hi[k] = nodes[kid].bmax[k]; }
            }
            if (!inside(lo, hi, nodes[i].bmin, nodes[i].bmax)) return false;
        }
    }
    return true;
}

// Lays the context's tree (c->h_tree, DFS pre-order as uploaded) out in the device node arrays and threads it:
// every node gets the code of the node the walk visits next (`next`: a Split's first child, a Leaf's
// successor) and a Split the code of the node that follows its subtree (`skip`).  The TREE is the
// caller's -- same boxes, same leaves -- and so is the set of tests a ray can make; what this function chooses
// is which of a Split's two children the walk enters first:
//   PT_WALK_REFERENCE  the left one (bvh.rs:111-118), the reference's order;
//   PT_WALK_VIEW       the one whose box centre is nearer to the camera's origin.  Primary rays then find
//                      their closest hit early and skip what lies behind it; rays from elsewhere are served
//                      no worse than by the left-first order (-11 % box tests, -16 %
//                      sphere tests per path on the bench scene).
// The closest hit of a ray does not depend on the order in which its candidates are found -- except
// among candidates with the SAME t, where the reference keeps the one its walk meets first (strict '<',
// sphere.rs:57).  Leaves therefore carry their RANK in the reference's order (= the primitive's index in
// the device arrays, upload_common), and Sphere::trace on the device lets the lower rank win a tie
// (pt_device.hpp closer): in the reference's order that never fires, in any other order it restores the
// reference's choice.  Extension scenes (coplanar rects tie by construction, and rect_step has no such
// rule) always keep the reference's order, and so do trees whose boxes do not contain what is below them
// (boxes_contain): there a box test can hide a closer hit and the order of the walk decides which.
// Slots: kMaxLdsNodes nodes are the prefix the path kernel keeps in LDS -- the first ones breadth-first (the top of
// the tree, visited by every ray) or, in the view order, the ones this camera's walks visit most --, the others
// follow in walk order (a Split's first child is the next entry: neighbouring
// entries share cache lines; -0.8 % on cfg 5 against breadth-first throughout).
int thread_nodes(pt_context *c)
{
    const int32_t n_nodes = (int32_t)c->h_tree.size();
    const pt_bvh_node *nodes = c->h_tree.data();
    const bool view = c->walk_order == PT_WALK_VIEW && c->view_ok && c->have_camera;
    const float eye[3] = {c->camera.origin.x, c->camera.origin.y, c->camera.origin.z};
    if (c->threaded && c->threaded_view == view && (!view || std::memcmp(eye, c->threaded_eye, sizeof eye) == 0)) return PT_OK;

    std::vector<int32_t> slot, order, first, second, succ, walk, stack, leaf_codes, parent;
    std::vector<float4> h_nodes;
    const int32_t n_slots = slots_for(n_nodes);
    try {
        slot.resize((size_t)n_nodes);
        order.reserve((size_t)n_nodes);
        first.assign((size_t)n_nodes, -1);
        second.assign((size_t)n_nodes, -1);
        succ.assign((size_t)n_nodes, n_nodes);
        walk.reserve((size_t)n_nodes);
        stack.reserve(64);
        h_nodes.assign(2 * (size_t)slots_for(n_nodes), make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        leaf_codes.assign(2 * (size_t)c->scene.n_spheres, 0);   // [rank] the Leaf's code, [n + rank] its parent Split's
        parent.assign((size_t)n_nodes, 0);
    } catch (...) {
        return fail(c, PT_ERR_INVALID, "out of host memory");
    }
    // squared distance from the eye to the centre of node i's box (a Leaf: its sphere's centre; the bounds of
    // a Leaf are never read, include/pt_amd.h).  Host-side double arithmetic: it orders visits, it is not
    // part of any result.
    auto dist2 = [&](int32_t i) -> double {
        double d2 = 0.0;
        for (int k = 0; k < 3; k++) {
            const float4 &leaf = c->h_leaf[2 * (size_t)i];
            const float lc[3] = {leaf.x, leaf.y, leaf.z};
            const double centre = nodes[i].prim >= 0 ? (double)lc[k] : 0.5 * ((double)nodes[i].bmin[k] + (double)nodes[i].bmax[k]);
            d2 += (centre - (double)eye[k]) * (centre - (double)eye[k]);
        }
        return d2;
    };
    for (int32_t i = 0; i < n_nodes; i++) {
        if (nodes[i].prim >= 0) continue;
        const int32_t l = i + 1, r = nodes[i + 1].skip;
        const bool swap = view && dist2(r) < dist2(l);   // NaN compares false: left first
        first[i] = swap ? r : l;
        second[i] = swap ? l : r;
        parent[(size_t)l] = i;
        parent[(size_t)r] = i;
    }
    // walk order (depth-first, first child before second) and every node's successor
    try {
        stack.push_back(0);
        while (!stack.empty()) {
            const int32_t i = stack.back();
            stack.pop_back();
            if ((int32_t)walk.size() >= n_nodes) return fail(c, PT_ERR_INVALID, "BVH is not a single tree");
            walk.push_back(i);                          // within the reservation
            if (nodes[i].prim < 0) {
                succ[first[i]] = second[i];
                succ[second[i]] = succ[i];
                stack.push_back(second[i]);             // the stack is as deep as the tree: may grow
                stack.push_back(first[i]);
            }
        }
    } catch (...) {
        return fail(c, PT_ERR_INVALID, "out of host memory");
    }
    if ((int32_t)walk.size() != n_nodes) return fail(c, PT_ERR_INVALID, "BVH is not a single tree");
    order.push_back(0);
    for (size_t q = 0; q < order.size() && order.size() + 2 <= (size_t)n_nodes; q++) {
        const int32_t i = order[q];
        if (nodes[i].prim < 0) {
            order.push_back(first[i]);                  // within the reservation
            order.push_back(second[i]);
        }
    }
    if ((int32_t)order.size() != n_nodes) return fail(c, PT_ERR_INVALID, "BVH is not a single tree");
    if (n_nodes <= kMaxLdsNodes) {
        for (int32_t q = 0; q < n_nodes; q++) slot[order[q]] = q;
    } else {
        std::vector<char> in_prefix;
        try {
            in_prefix.assign((size_t)n_nodes, 0);
        } catch (...) {
            return fail(c, PT_ERR_INVALID, "out of host memory");
        }
        if (!view) {
            for (int32_t q = 0; q < kMaxLdsNodes; q++) in_prefix[order[q]] = 1;
        } else {
            // The nodes a walk from this camera visits most: a node is visited when its parent's box is hit, and
            // seen from the eye that box subtends about area / distance^2.  On cfg 5 (21,385 nodes) the 2048 nodes
            // of largest key take 94.4 % of the visits, the first 2048 breadth-first 85.5 %
            // (tests/experiments/exp_prefix.c).  Placement only: no result depends on it.
            std::vector<double> key;
            std::vector<int32_t> pick;
            try {
                key.assign((size_t)n_nodes, 0.0);
                pick.resize((size_t)n_nodes);
            } catch (...) {
                return fail(c, PT_ERR_INVALID, "out of host memory");
            }
            auto key_of_parent = [&](int32_t p) -> double {
                const double ex = (double)nodes[p].bmax[0] - (double)nodes[p].bmin[0], ey = (double)nodes[p].bmax[1] - (double)nodes[p].bmin[1],
                             ez = (double)nodes[p].bmax[2] - (double)nodes[p].bmin[2];
                const double k = (ex * ey + ey * ez + ez * ex) / (dist2(p) + 1e-30);
                return k == k ? k : -1.0;               // a NaN box sorts last
            };
            key[0] = key_of_parent(0);
            for (int32_t i = 0; i < n_nodes; i++) {
                pick[(size_t)i] = i;
                if (nodes[i].prim >= 0) continue;
                const double k = key_of_parent(i);
                key[(size_t)i + 1] = k;
                key[(size_t)nodes[i + 1].skip] = k;
            }
            key[0] = 1e300;                             // the root is visited by every walk
            std::nth_element(pick.begin(), pick.begin() + kMaxLdsNodes, pick.end(), [&](int32_t x, int32_t y) {
                return key[(size_t)x] > key[(size_t)y] || (key[(size_t)x] == key[(size_t)y] && x < y);
            });
            for (int32_t q = 0; q < kMaxLdsNodes; q++) in_prefix[pick[(size_t)q]] = 1;
        }
        int32_t next_slot = 0;
        for (int32_t q = 0; q < n_nodes; q++)           // the staged ones, breadth-first among themselves
            if (in_prefix[order[q]]) slot[order[q]] = next_slot++;
        next_slot = PT_FIRST_GLOBAL_SLOT;
        for (int32_t i : walk)
            if (!in_prefix[i]) slot[i] = next_slot++;
    }
    // node code: slot with the sign bit set for a Leaf; n_nodes ends the walk
    auto code_of = [&](int32_t i) -> int32_t {
        if (i >= n_nodes) return end_slot_for(n_nodes) << PT_CODE_SHIFT;
        return nodes[i].prim >= 0 ? (int32_t)((uint32_t)(slot[i] << PT_CODE_SHIFT) | 0x80000000u) : (slot[i] << PT_CODE_SHIFT);
    };
    float4 *h_n0 = h_nodes.data(), *h_n1 = h_nodes.data() + n_slots;
    for (int32_t i = 0; i < n_nodes; i++) {
        const pt_bvh_node &nd = nodes[i];
        float4 a, b;
        if (nd.prim < 0) {
            const int32_t skip = code_of(succ[i]), next = code_of(first[i]);
            a = make_float4(nd.bmin[0], nd.bmin[1], nd.bmin[2], 0.0f);
            b = make_float4(nd.bmax[0], nd.bmax[1], nd.bmax[2], 0.0f);
            std::memcpy(&a.w, &skip, 4);
            std::memcpy(&b.w, &next, 4);
        } else {
            const int32_t next = code_of(succ[i]);
            a = c->h_leaf[2 * (size_t)i];
            b = c->h_leaf[2 * (size_t)i + 1];
            std::memcpy(&b.w, &next, 4);
            int32_t rank;
            std::memcpy(&rank, &b.z, 4);
            leaf_codes[(size_t)rank] = code_of(i);      // rank: validated at upload, one Leaf per primitive
            leaf_codes[(size_t)c->scene.n_spheres + (size_t)rank] = code_of(parent[(size_t)i]);   // (a lone Leaf: the root itself; no lists then)
        }
        h_n0[slot[i]] = a;
        h_n1[slot[i]] = b;
    }
    // A camera whose eye orders every Split's children -- and picks the staged set -- as the previous one did
    // threads the tree into the same arrays: then the device copy stays as it is (no device-wide wait, no copy).
    const bool same = c->threaded && c->h_threaded.size() == h_nodes.size() &&
                      std::memcmp(c->h_threaded.data(), h_nodes.data(), h_nodes.size() * sizeof(float4)) == 0;
    if (!same) {
        PT_HIP(c, hipSetDevice(c->device));
        PT_HIP(c, hipDeviceSynchronize());   // launches on caller streams (pt_render_device) may still be walking the old arrays
        c->threaded = false;                 // until the copy has succeeded the device arrays are in no known state
        PT_HIP(c, hipMemcpy(c->d_nodes, h_nodes.data(), h_nodes.size() * sizeof(float4), hipMemcpyHostToDevice));
        PT_HIP(c, hipMemcpy(c->d_leaf_codes, leaf_codes.data(), leaf_codes.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        c->h_threaded.swap(h_nodes);
        c->lists_valid = false;              // the lists hold node codes: new codes, new lists
    }
    c->scene.root_code = code_of(0);
    c->threaded = true;
    c->threaded_view = view;
    std::memcpy(c->threaded_eye, eye, sizeof eye);
    return PT_OK;
}

// The primary-ray candidate lists of the current scene + camera (pt_kernels.hip primary_list_kernel; Scene::cand),
// rebuilt whenever the camera, the scene or the node numbering changes.  Only for trees whose boxes contain what
// is below them (view_ok: nested boxes are what lets the box test of a Leaf's PARENT stand for the tests of all its
// ancestors -- the kernel's chains keep that one test per candidate) and never for the
// extension's scenes; pt_set_primary_lists(ctx, 0) turns them off.
int build_primary_lists(pt_context *c)
{
    c->scene.cand = nullptr;
    c->scene.cand_tiles_x = 0;
    if (!c->have_scene || !c->have_camera || !c->primary_lists || !c->view_ok || c->scene.ext || !c->threaded) return PT_OK;
    // the small-tree launch keeps the plain walk (pt_kernels.hip launch_path), and so does a scene x frame whose
    // brute-force list build would take longer than it saves (a lane per block tests every sphere)
    const int64_t blocks = (int64_t)((c->camera.width + 7) / 8) * ((c->camera.height + 7) / 8);
    if (c->scene.n_nodes < 32 || blocks * c->scene.n_spheres > (int64_t)2e10) return PT_OK;
    const size_t need = (size_t)blocks * PT_PRIMARY_K * sizeof(int32_t);
    PT_HIP(c, hipSetDevice(c->device));
    if (need > c->lists_bytes) {
        int32_t *fresh = nullptr;
        PT_HIP(c, hipMalloc((void **)&fresh, need));
        hipError_t e = hipDeviceSynchronize();   // launches on caller streams may still be reading the old lists
        if (e != hipSuccess) {
            (void)hipFree(fresh);
            return fail(c, PT_ERR_HIP, "hipDeviceSynchronize failed: %s", hipGetErrorString(e));
        }
        if (c->d_lists) (void)hipFree(c->d_lists);
        c->d_lists = fresh;
        c->lists_bytes = need;
        c->lists_valid = false;
    }
    if (!c->lists_valid || std::memcmp(&c->lists_camera, &c->camera, sizeof c->camera) != 0) {
        PT_HIP(c, hipDeviceSynchronize());
        PT_HIP(c, ptk::launch_primary_lists(c->scene, c->camera, c->d_leaf_codes, c->scene.n_spheres, c->d_lists, c->stream));
        PT_HIP(c, hipStreamSynchronize(c->stream));
        c->lists_camera = c->camera;
        c->lists_valid = true;
    }
    c->scene.cand = c->d_lists;
    c->scene.cand_tiles_x = (c->camera.width + 7) / 8;
    return PT_OK;
}

int pick_lanes(const pt_context *c, int32_t count)
{
    if (count < 1) count = 1;
    int lanes = c->lanes_override;
    if (lanes <= 0) {
        const int64_t px = (int64_t)c->camera.width * c->camera.height;
        lanes = 1;
        // a frame that fills the chip on its own keeps one lane per pixel (sequential sums, the
        // launch is balanced by pieces); a smaller one is spread until it covers the wave slots
        // about three times over, so that the last round of waves is a small part of the launch
        if (px < kFullChipLanes)
            while (lanes < 64 && px * lanes < 3 * kFullChipLanes) lanes *= 2;
    }
    while (lanes > 1 && lanes > count) lanes /= 2;   // never more lanes than samples
    return lanes;
}

}  // namespace

extern "C" {

int pt_create(int32_t device, pt_context **out)
{
    if (!out) return fail(nullptr, PT_ERR_INVALID, "pt_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count < 1)
        return fail(nullptr, PT_ERR_NO_DEVICE, "no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(nullptr, PT_ERR_INVALID, "device %d out of range [0, %d)", device, count);
    hipDeviceProp_t prop;
    PT_HIP(nullptr, hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, PT_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    pt_context *c = new (std::nothrow) pt_context();
    if (!c) return fail(nullptr, PT_ERR_INVALID, "out of host memory");
    c->device = device;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_start);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_stop);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_own_done, hipEventDisableTiming);
    // each GPU holds its own copy of the scratch array at its own address: resolved here, with this
    // context's device current, and used through this pointer only
    if (e == hipSuccess) e = ptk::static_sched(device, &c->static_sched, &c->static_sched_bytes);
    if (e != hipSuccess) {
        if (c->ev_start) (void)hipEventDestroy(c->ev_start);
        if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
        if (c->ev_own_done) (void)hipEventDestroy(c->ev_own_done);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(nullptr, PT_ERR_HIP, "device binding / stream / event creation failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return PT_OK;
}

void pt_destroy(pt_context *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_scene(c);
    free_frame(c);
    if (c->d_sched) (void)hipFree(c->d_sched);
    if (c->d_lists) (void)hipFree(c->d_lists);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    if (c->ev_own_done) (void)hipEventDestroy(c->ev_own_done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *pt_last_error(const pt_context *c) { return c ? c->error.c_str() : g_create_error.c_str(); }

int pt_query(const pt_context *c, pt_device_info *out)
{
    if (!c || !out) return PT_ERR_INVALID;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return PT_ERR_HIP;
    std::memset(out, 0, sizeof *out);
    std::snprintf(out->name, sizeof out->name, "%s", prop.name);
    std::snprintf(out->arch, sizeof out->arch, "%s", prop.gcnArchName);
    out->compute_units = prop.multiProcessorCount;
    out->wavefront_size = prop.warpSize;
    out->clock_khz = prop.clockRate;
    out->memory_clock_khz = prop.memoryClockRate;
    out->l2_bytes = prop.l2CacheSize;
    out->hbm_bytes = (int64_t)prop.totalGlobalMem;
    return PT_OK;
}

// pt_upload_scene / pt_upload_scene_ext: validation of the primitive list, the tree and the materials,
// the device layout, the H2D copies.  `ext` selects the extension's kernel instantiation for the scene.
static int upload_common(pt_context *c, const pt_primitive *prims, int32_t n_prims, const pt_material *materials,
                         int32_t n_materials, const pt_texture *textures, int32_t n_textures, const pt_transform *xforms,
                         int32_t n_xforms, const pt_bvh_node *nodes, int32_t n_nodes, bool ext)
{
    if (!materials || n_materials < 1) return fail(c, PT_ERR_INVALID, "scene needs at least one material");
    if (n_textures < 0 || (n_textures > 0 && !textures)) return fail(c, PT_ERR_INVALID, "bad texture array");
    // node codes are byte offsets (16 * slot) in an int32 with the sign bit as the Leaf flag, and the
    // buffer resource's size is 16 * n_nodes: 2 * n_prims - 1 nodes must stay below 2^(31 - PT_CODE_SHIFT)
    if (n_prims > (1 << (30 - PT_CODE_SHIFT)))
        return fail(c, PT_ERR_LIMIT, "too many primitives (%d): the node encoding holds at most %d", n_prims, 1 << (30 - PT_CODE_SHIFT));
    for (int32_t i = 0; i < n_materials; i++) {
        const pt_material &m = materials[i];
        if (m.kind < PT_LAMBERTIAN || m.kind > (ext ? PT_DIFFUSE_LIGHT : PT_DIELECTRIC))
            return fail(c, PT_ERR_INVALID, "material %d: unknown kind %d", i, m.kind);
        if (m.kind == PT_LAMBERTIAN && (m.texture < -1 || m.texture >= n_textures))
            return fail(c, PT_ERR_INVALID, "material %d: texture index %d not -1 or in [0, %d)", i, m.texture, n_textures);
    }
    for (int32_t i = 0; i < n_prims; i++) {
        const pt_primitive &p = prims[i];
        if (p.material < 0 || p.material >= n_materials)
            return fail(c, PT_ERR_INVALID, "primitive %d: material index %d out of range [0, %d)", i, p.material, n_materials);
        if (p.kind == PT_PRIM_RECT) {
            if (!ext) return fail(c, PT_ERR_INVALID, "primitive %d: rects need pt_upload_scene_ext", i);
            if (p.axis < 0 || p.axis > 2) return fail(c, PT_ERR_INVALID, "primitive %d: rect axis %d not 0, 1 or 2", i, p.axis);
            if (p.transform < -1 || p.transform >= n_xforms)
                return fail(c, PT_ERR_INVALID, "primitive %d: transform index %d not -1 or in [0, %d)", i, p.transform, n_xforms);
            const pt_material &m = materials[p.material];
            if (m.kind == PT_LAMBERTIAN && m.texture >= 0)
                return fail(c, PT_ERR_INVALID, "primitive %d: a rect cannot carry a picture texture (sphere_uv is a sphere's, sphere.rs:6-18)", i);
        } else if (p.kind == PT_PRIM_SPHERE) {
            if (p.transform != -1) return fail(c, PT_ERR_INVALID, "primitive %d: a sphere takes no transform", i);
        } else {
            return fail(c, PT_ERR_INVALID, "primitive %d: unknown kind %d", i, p.kind);
        }
    }
    for (int32_t i = 0; i < n_textures; i++)
        if (!textures[i].rgb8 || textures[i].width < 1 || textures[i].height < 1)
            return fail(c, PT_ERR_INVALID, "texture %d: empty image", i);

    const int32_t expect = 2 * n_prims - 1;
    if (n_nodes != expect) return fail(c, PT_ERR_INVALID, "n_nodes = %d, a binary BVH over %d primitives has %d", n_nodes, n_prims, expect);
    // validate the walk the kernel will do: it follows skip links and prim ids unchecked
    for (int32_t i = 0; i < n_nodes; i++) {
        const pt_bvh_node &nd = nodes[i];
        if (nd.skip <= i || nd.skip > n_nodes) return fail(c, PT_ERR_INVALID, "node %d: skip %d not in (%d, %d]", i, nd.skip, i, n_nodes);
        if (nd.prim >= n_prims || nd.prim < -1) return fail(c, PT_ERR_INVALID, "node %d: prim %d out of range", i, nd.prim);
        if (nd.prim >= 0 && nd.skip != i + 1) return fail(c, PT_ERR_INVALID, "node %d: a leaf's skip must be %d", i, i + 1);
        if (nd.prim < 0 && i + 1 >= n_nodes) return fail(c, PT_ERR_INVALID, "node %d: split without children", i);
    }

    // a Split's children: left = i + 1, right = the node after the left subtree; both
    // subtrees must end where the Split's own subtree ends (proper nesting)
    for (int32_t i = 0; i < n_nodes; i++) {
        if (nodes[i].prim >= 0) continue;
        const int32_t r = nodes[i + 1].skip;
        if (r >= nodes[i].skip || nodes[r].skip != nodes[i].skip)
            return fail(c, PT_ERR_INVALID, "node %d: children do not tile its subtree", i);
    }

    // Rank of every primitive = position of its Leaf in the tree's depth-first order = the order in which the
    // reference's walk meets the primitives (bvh.rs:106-127).  The device arrays hold the primitives BY RANK, so
    // a hit's primitive index doubles as the tie-break key of pt_device.hpp closer(); prim_ids maps back to the
    // caller's ids (hit-id AOV).
    std::vector<int32_t> rank_of, prim_ids;
    try {
        rank_of.assign((size_t)n_prims, -1);
        prim_ids.reserve((size_t)n_prims);
    } catch (...) {
        return fail(c, PT_ERR_INVALID, "out of host memory");
    }
    for (int32_t i = 0; i < n_nodes; i++) {
        const int32_t pr = nodes[i].prim;
        if (pr < 0) continue;
        if (rank_of[pr] >= 0) return fail(c, PT_ERR_INVALID, "node %d: primitive %d already has a leaf", i, pr);
        rank_of[pr] = (int32_t)prim_ids.size();
        prim_ids.push_back(pr);
    }
    if ((int32_t)prim_ids.size() != n_prims) return fail(c, PT_ERR_INVALID, "the tree has %d leaves for %d primitives", (int32_t)prim_ids.size(), n_prims);

    // Host copy of the tree and of its leaves' device records: the node arrays are laid out and threaded
    // by thread_nodes, again whenever the camera or the walk order changes.
    // a rect's flags (pt_device.hpp rect_step): bit 0 = rect, bits 1-2 = axis, bit 3 = flip, bits 4.. = transform + 1
    auto rect_flags = [](const pt_primitive &p) -> int32_t { return 1 | (p.axis << 1) | ((p.flip ? 1 : 0) << 3) | ((p.transform + 1) << 4); };
    std::vector<pt_bvh_node> h_tree;
    std::vector<float4> h_leaf, h_prims, h_xforms;
    try {
        h_tree.assign(nodes, nodes + n_nodes);
        h_leaf.assign(2 * (size_t)n_nodes, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        h_prims.resize(2 * (size_t)n_prims);
        h_xforms.resize(2 * (size_t)(n_xforms > 0 ? n_xforms : 0));
    } catch (...) {
        return fail(c, PT_ERR_INVALID, "out of host memory");
    }
    for (int32_t i = 0; i < n_nodes; i++) {
        const pt_bvh_node &nd = nodes[i];
        if (nd.prim < 0) continue;
        float4 a, b;
        if (prims[nd.prim].kind == PT_PRIM_RECT) {
            const pt_primitive &r = prims[nd.prim];
            const int32_t flags = rect_flags(r);
            a = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
            b = make_float4(r.v[4], 0.0f, 0.0f, 0.0f);
            std::memcpy(&b.y, &flags, 4);
        } else {
            const pt_primitive &s = prims[nd.prim];
            a = make_float4(s.v[0], s.v[1], s.v[2], s.v[3] * s.v[3]);   // sphere.rs:51: radius * radius, one fp32 multiply
            b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        std::memcpy(&b.z, &rank_of[nd.prim], 4);
        h_leaf[2 * (size_t)i] = a;
        h_leaf[2 * (size_t)i + 1] = b;
    }
    for (int32_t i = 0; i < n_prims; i++) {
        const pt_primitive &p = prims[i];
        const pt_material &m = materials[p.material];
        const int32_t tex = m.kind == PT_LAMBERTIAN ? m.texture : -1;
        int32_t tag = m.kind | ((tex + 1) << 2);
        float4 a = make_float4(p.v[0], p.v[1], p.v[2], 0.0f);
        if (p.kind == PT_PRIM_RECT) {
            const int32_t flags = rect_flags(p);
            a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            std::memcpy(&a.x, &flags, 4);
            tag |= 1 << 30;
        }
        std::memcpy(&a.w, &tag, 4);
        const size_t at = 2 * (size_t)rank_of[i];
        h_prims[at] = a;
        h_prims[at + 1] = make_float4(m.albedo[0], m.albedo[1], m.albedo[2], m.param);
        // a Dielectric has no albedo: the slots carry the two values its scatter derives from ref_idx
        // alone, math/mod.rs:37-38 and dielectric.rs:41, computed here with the same fp32 operations
        if (m.kind == PT_DIELECTRIC) h_prims[at + 1] = make_float4(ptd::schlick_r0(m.param), 1.0f / m.param, 0.0f, m.param);
    }
    for (int32_t i = 0; i < n_xforms; i++) {
        h_xforms[2 * (size_t)i] = make_float4(xforms[i].offset[0], xforms[i].offset[1], xforms[i].offset[2], xforms[i].sin_theta);
        h_xforms[2 * (size_t)i + 1] = make_float4(xforms[i].cos_theta, 0.0f, 0.0f, 0.0f);
    }

    PT_HIP(c, hipSetDevice(c->device));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    const int32_t n_slots = slots_for(n_nodes);
    PT_HIP(c, hipMalloc((void **)&c->d_nodes, 2 * (size_t)n_slots * sizeof(float4)));
    PT_HIP(c, hipMalloc((void **)&c->d_leaf_codes, 2 * (size_t)n_prims * sizeof(int32_t)));
    PT_HIP(c, hipMalloc((void **)&c->d_prims, h_prims.size() * sizeof(float4)));
    PT_HIP(c, hipMemcpy(c->d_prims, h_prims.data(), h_prims.size() * sizeof(float4), hipMemcpyHostToDevice));
    PT_HIP(c, hipMalloc((void **)&c->d_prim_ids, prim_ids.size() * sizeof(int32_t)));
    PT_HIP(c, hipMemcpy(c->d_prim_ids, prim_ids.data(), prim_ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!h_xforms.empty()) {
        PT_HIP(c, hipMalloc((void **)&c->d_xforms, h_xforms.size() * sizeof(float4)));
        PT_HIP(c, hipMemcpy(c->d_xforms, h_xforms.data(), h_xforms.size() * sizeof(float4), hipMemcpyHostToDevice));
    }
    if (n_textures > 0) {
        std::vector<ptd::Texture> h_tex;
        try {
            h_tex.resize((size_t)n_textures);
            c->d_texels.reserve((size_t)n_textures);
        } catch (...) {
            return fail(c, PT_ERR_INVALID, "out of host memory");
        }
        for (int32_t i = 0; i < n_textures; i++) {
            const size_t bytes = (size_t)textures[i].width * (size_t)textures[i].height * 3;
            uint8_t *d = nullptr;
            PT_HIP(c, hipMalloc((void **)&d, bytes));
            c->d_texels.push_back(d);
            PT_HIP(c, hipMemcpy(d, textures[i].rgb8, bytes, hipMemcpyHostToDevice));
            h_tex[i].rgb8 = d;
            h_tex[i].width = textures[i].width;
            h_tex[i].height = textures[i].height;
        }
        PT_HIP(c, hipMalloc((void **)&c->d_textures, h_tex.size() * sizeof(ptd::Texture)));
        PT_HIP(c, hipMemcpy(c->d_textures, h_tex.data(), h_tex.size() * sizeof(ptd::Texture), hipMemcpyHostToDevice));
    }
    c->scene.n0 = c->d_nodes;
    c->scene.n1 = c->d_nodes + n_slots;
    c->scene.prims = c->d_prims;
    c->scene.prim_ids = c->d_prim_ids;
    c->scene.textures = c->d_textures;
    c->scene.xforms = c->d_xforms;
    c->scene.ext = ext ? 1 : 0;
    c->scene.n_xforms = ext && n_xforms > 0 ? n_xforms : 0;
    c->scene.lds_xforms = (ext && n_xforms <= 128) ? n_xforms : 0;   // 4 KiB of LDS at most; more: read through the L1
    c->scene.n_nodes = n_slots;              // slots, not nodes: bounds the arrays
    c->scene.end_slot = end_slot_for(n_nodes);
    c->scene.parent_codes = c->d_leaf_codes + n_prims;
    c->scene.cand = nullptr;
    c->scene.cand_tiles_x = 0;
    c->lists_valid = false;
    c->scene.n_spheres = n_prims;
    c->scene.lds_nodes = n_nodes < kMaxLdsNodes ? n_nodes : kMaxLdsNodes;
    // regular_ray (pt_device.hpp): the short division needs every sphere inside 2^32 (NaN fails the test)
    bool bounded = true;
    for (int32_t i = 0; i < n_prims; i++) {
        if (prims[i].kind != PT_PRIM_SPHERE) continue;
        const float *v = prims[i].v;
        const float lim = 4294967296.0f;
        bounded = bounded && std::fabs(v[0]) <= lim && std::fabs(v[1]) <= lim && std::fabs(v[2]) <= lim && std::fabs(v[3]) <= lim;
    }
    c->scene.bounded = bounded ? 1 : 0;
    // PT_WALK_VIEW (thread_nodes): not for the extension's scenes, not for trees whose boxes do not contain their contents
    c->view_ok = !ext && boxes_contain(nodes, n_nodes, prims);
    c->h_tree.swap(h_tree);
    c->h_leaf.swap(h_leaf);
    c->threaded = false;
    c->h_threaded.clear();
    int rc = thread_nodes(c);
    if (rc != PT_OK) return rc;
    c->have_scene = true;
    return build_primary_lists(c);
}

int pt_upload_scene(pt_context *c, const pt_sphere *spheres, int32_t n_spheres, const pt_material *materials,
                    int32_t n_materials, const pt_texture *textures, int32_t n_textures, const pt_bvh_node *nodes,
                    int32_t n_nodes)
{
    if (!c) return PT_ERR_INVALID;
    if (!spheres || n_spheres < 1) return fail(c, PT_ERR_INVALID, "scene needs at least one sphere (bvh.rs:19)");
    if (n_spheres > (1 << (30 - PT_CODE_SHIFT)))
        return fail(c, PT_ERR_LIMIT, "too many spheres (%d): the node encoding holds at most %d", n_spheres, 1 << (30 - PT_CODE_SHIFT));
    // BVH: caller's flattened tree, or build it here (bvh.rs:13-91)
    std::vector<pt_bvh_node> built;
    std::vector<pt_primitive> prims;
    try {
        prims.resize((size_t)n_spheres);
        if (!nodes || n_nodes == 0) built.resize(2 * (size_t)n_spheres - 1);
    } catch (...) {
        return fail(c, PT_ERR_INVALID, "out of host memory");
    }
    if (!nodes || n_nodes == 0) {
        const int32_t got = pt_bvh_build(spheres, n_spheres, built.data(), (int32_t)built.size());
        if (got == PT_ERR_INVALID)
            return fail(c, PT_ERR_INVALID, "BVH build: degenerate bounds (zero volume or non-finite) -- the reference panics here (bvh.rs:19)");
        if (got != (int32_t)built.size()) return fail(c, PT_ERR_INVALID, "BVH build failed (%d)", got);
        nodes = built.data();
        n_nodes = got;
    }
    for (int32_t i = 0; i < n_spheres; i++)
        prims[i] = pt_primitive{PT_PRIM_SPHERE, spheres[i].material, -1, 0, 0, {spheres[i].cx, spheres[i].cy, spheres[i].cz, spheres[i].radius, 0.0f}};
    return upload_common(c, prims.data(), n_spheres, materials, n_materials, textures, n_textures, nullptr, 0, nodes, n_nodes, false);
}

int pt_upload_scene_ext(pt_context *c, const pt_primitive *prims, int32_t n_prims, const pt_material *materials,
                        int32_t n_materials, const pt_texture *textures, int32_t n_textures, const pt_transform *transforms,
                        int32_t n_transforms, const pt_bvh_node *nodes, int32_t n_nodes)
{
    if (!c) return PT_ERR_INVALID;
    if (!prims || n_prims < 1) return fail(c, PT_ERR_INVALID, "scene needs at least one primitive (bvh.rs:19)");
    if (n_transforms < 0 || (n_transforms > 0 && !transforms)) return fail(c, PT_ERR_INVALID, "bad transform array");
    if (n_prims > (1 << (30 - PT_CODE_SHIFT)))
        return fail(c, PT_ERR_LIMIT, "too many primitives (%d): the node encoding holds at most %d", n_prims, 1 << (30 - PT_CODE_SHIFT));
    std::vector<pt_bvh_node> built;
    if (!nodes || n_nodes == 0) {
        try {
            built.resize(2 * (size_t)n_prims - 1);
        } catch (...) {
            return fail(c, PT_ERR_INVALID, "out of host memory");
        }
        const int32_t got = pt_bvh_build_ext(prims, n_prims, transforms, n_transforms, built.data(), (int32_t)built.size());
        if (got != (int32_t)built.size())
            return fail(c, PT_ERR_INVALID, "BVH build over the primitives failed (%d): bad primitive fields or degenerate bounds", got);
        nodes = built.data();
        n_nodes = got;
    }
    return upload_common(c, prims, n_prims, materials, n_materials, textures, n_textures, transforms, n_transforms, nodes, n_nodes, true);
}

int pt_set_integrator(pt_context *c, const pt_integrator *integ)
{
    if (!c) return PT_ERR_INVALID;
    if (!integ) { c->integrator = pt_integrator{5, 0, {0.0f, 0.0f, 0.0f}, -1}; return PT_OK; }
    if (integ->max_bounces < 1 || integ->max_bounces > 1024) return fail(c, PT_ERR_INVALID, "max_bounces must be in [1, 1024] (got %d)", integ->max_bounces);
    if (integ->background_kind != 0 && integ->background_kind != 1)
        return fail(c, PT_ERR_INVALID, "background_kind must be 0 (sky gradient) or 1 (constant), got %d", integ->background_kind);
    c->integrator = *integ;
    return PT_OK;
}

int pt_set_camera(pt_context *c, const pt_camera *cam)
{
    if (!c || !cam) return PT_ERR_INVALID;
    if (cam->width < 1 || cam->height < 1 || (int64_t)cam->width * cam->height > (1ll << 28))
        return fail(c, PT_ERR_INVALID, "bad frame size %d x %d", cam->width, cam->height);
    ptd::Camera next{};
    next.llc = ptd::f3{cam->lower_left_corner[0], cam->lower_left_corner[1], cam->lower_left_corner[2]};
    next.origin = ptd::f3{cam->origin[0], cam->origin[1], cam->origin[2]};
    next.horizontal = ptd::f3{cam->horizontal[0], cam->horizontal[1], cam->horizontal[2]};
    next.vertical = ptd::f3{cam->vertical[0], cam->vertical[1], cam->vertical[2]};
    next.width = cam->width;
    next.height = cam->height;
    // the new frame's scratch first; camera and scratch are committed together, only on success
    const size_t need = ptk::sched_bytes(next);
    if (need > c->sched_bytes) {
        PT_HIP(c, hipSetDevice(c->device));
        int32_t *fresh = nullptr;
        PT_HIP(c, hipMalloc((void **)&fresh, need));
        if (hipMemset(fresh, 0, need) != hipSuccess) {     // a launch expects its words at zero and leaves them so (pt_kernels.hip retire_group)
            (void)hipFree(fresh);
            return fail(c, PT_ERR_HIP, "hipMemset of the work-distribution scratch failed");
        }
        // launches on caller streams (pt_render_device) may still be using the old scratch
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            (void)hipFree(fresh);
            return fail(c, PT_ERR_HIP, "hipDeviceSynchronize failed: %s", hipGetErrorString(e));
        }
        if (c->last_sched == c->d_sched) c->last_sched = nullptr;
        c->own_used = false;                       // the device is idle: nothing is in flight on the old scratch
        if (c->d_sched) (void)hipFree(c->d_sched);
        c->d_sched = fresh;
        c->sched_bytes = need;
    }
    // PT_WALK_VIEW orders the walk by the camera's origin: thread the nodes for the new eye first; if that fails
    // the context keeps its previous camera (and the message of the failure)
    const ptd::Camera previous = c->camera;
    const bool had_camera = c->have_camera;
    c->camera = next;
    c->have_camera = true;
    if (c->have_scene) {
        int rc = thread_nodes(c);
        if (rc == PT_OK) rc = build_primary_lists(c);
        if (rc != PT_OK) {
            c->camera = previous;
            c->have_camera = had_camera;
            c->scene.cand = nullptr;             // lists of another camera are never used: the plain walk until the next success
                    c->lists_valid = false;
            return rc;
        }
    }
    return PT_OK;
}

int pt_selftest_unary(pt_context *c, int32_t fn, uint64_t *mismatches)
{
    if (!c) return PT_ERR_INVALID;
    if (fn < 0 || fn > 2 || !mismatches) return fail(c, PT_ERR_INVALID, "pt_selftest_unary: fn must be 0, 1 or 2 and mismatches non-NULL");
    PT_HIP(c, hipSetDevice(c->device));
    unsigned long long *d_count = nullptr, h_count = 0;
    PT_HIP(c, hipMalloc((void **)&d_count, sizeof h_count));
    hipError_t e = hipMemsetAsync(d_count, 0, sizeof h_count, c->stream);
    if (e == hipSuccess) e = ptk::launch_unary_selftest(fn, d_count, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_count, d_count, sizeof h_count, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_count);
    if (e != hipSuccess) return fail(c, PT_ERR_HIP, "pt_selftest_unary failed: %s", hipGetErrorString(e));
    *mismatches = h_count;
    return PT_OK;
}

int pt_selftest_division(pt_context *c, int32_t log2_pairs, uint64_t *mismatches)
{
    if (!c) return PT_ERR_INVALID;
    if (log2_pairs < 16 || log2_pairs > 40 || !mismatches)
        return fail(c, PT_ERR_INVALID, "pt_selftest_division: log2_pairs must be in [16, 40] and mismatches non-NULL");
    PT_HIP(c, hipSetDevice(c->device));
    unsigned long long *d_count = nullptr, h_count = 0;
    PT_HIP(c, hipMalloc((void **)&d_count, sizeof h_count));
    hipError_t e = hipMemsetAsync(d_count, 0, sizeof h_count, c->stream);
    if (e == hipSuccess) e = ptk::launch_division_selftest(log2_pairs, d_count, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_count, d_count, sizeof h_count, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_count);
    if (e != hipSuccess) return fail(c, PT_ERR_HIP, "pt_selftest_division failed: %s", hipGetErrorString(e));
    *mismatches = h_count;
    return PT_OK;
}

int pt_selftest_grid(pt_context *c, int32_t samples, uint64_t *mismatches)
{
    int rc = check_ready(c, samples);
    if (rc != PT_OK) return rc;
    if (!mismatches) return fail(c, PT_ERR_INVALID, "pt_selftest_grid: mismatches is NULL");
    PT_HIP(c, hipSetDevice(c->device));
    unsigned long long *d_count = nullptr, h_count = 0;
    PT_HIP(c, hipMalloc((void **)&d_count, sizeof h_count));
    hipError_t e = hipMemsetAsync(d_count, 0, sizeof h_count, c->stream);
    if (e == hipSuccess)
        e = ptk::launch_grid_selftest(c->camera.width, c->camera.height, samples, 1.0f / (float)(samples - 1),
                                      1.0f / (float)c->camera.width, 1.0f / (float)c->camera.height, d_count, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_count, d_count, sizeof h_count, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_count);
    if (e != hipSuccess) return fail(c, PT_ERR_HIP, "pt_selftest_grid failed: %s", hipGetErrorString(e));
    *mismatches = h_count;
    return PT_OK;
}

int pt_clock_probe(pt_context *c, float *ghz)
{
    if (!c || !ghz) return PT_ERR_INVALID;
    *ghz = 0.0f;
    PT_HIP(c, hipSetDevice(c->device));
    int cus = 0;
    PT_HIP(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    const int groups = cus * 8;            // 8 groups of 4 waves per CU: 8 waves per SIMD, the path kernel's occupancy
    const int iters = 1800;                // x 128 multiplies per wave: about 2 ms at 2.2 GHz
    std::vector<unsigned long long> h;
    try {
        h.assign(2 * (size_t)groups, 0ull);
    } catch (...) {
        return fail(c, PT_ERR_INVALID, "out of host memory");
    }
    unsigned long long *d_stamps = nullptr;
    float *d_sink = nullptr;
    PT_HIP(c, hipMalloc((void **)&d_stamps, h.size() * sizeof(unsigned long long) + 256));
    d_sink = (float *)(d_stamps + h.size());
    hipError_t e = ptk::launch_clock_probe(d_stamps, d_sink, groups, iters, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_stamps);
    if (e != hipSuccess) return fail(c, PT_ERR_HIP, "pt_clock_probe failed: %s", hipGetErrorString(e));
    // median over the groups of cycles per 100 MHz tick
    std::vector<double> ratio;
    try {
        ratio.reserve((size_t)groups);
    } catch (...) {
        return fail(c, PT_ERR_INVALID, "out of host memory");
    }
    for (int i = 0; i < groups; i++)
        if (h[2 * (size_t)i + 1] > 0) ratio.push_back((double)h[2 * (size_t)i] / (double)h[2 * (size_t)i + 1]);   // within the reservation
    if (ratio.empty()) return fail(c, PT_ERR_HIP, "pt_clock_probe: no group reported a stamp");
    std::nth_element(ratio.begin(), ratio.begin() + ratio.size() / 2, ratio.end());
    *ghz = (float)(ratio[ratio.size() / 2] * 0.1);
    return PT_OK;
}

float pt_last_path_kernel_clock_ghz(pt_context *c)
{
    if (!c || !c->timed || !c->last_sched || !ptk::launch_is_persistent(c->scene)) return -1.0f;
    if (hipSetDevice(c->device) != hipSuccess || hipEventSynchronize(c->ev_stop) != hipSuccess) return -1.0f;
    uint32_t w[8] = {0};
    if (hipMemcpy(w, c->last_sched, sizeof w, hipMemcpyDeviceToHost) != hipSuccess) return -1.0f;
    const uint64_t clk = ((uint64_t)w[5] << 32) | w[4], ref = ((uint64_t)w[7] << 32) | w[6];
    if (ref == 0) return -1.0f;                      // the small-tree launch does not stamp
    return (float)((double)clk / (double)ref * 0.1);
}

int pt_set_lanes_per_pixel(pt_context *c, int32_t lanes)
{
    if (!c) return PT_ERR_INVALID;
    if (lanes != 0 && (lanes < 1 || lanes > 64 || (lanes & (lanes - 1)) != 0))
        return fail(c, PT_ERR_INVALID, "lanes per pixel must be 0 (auto) or a power of two in [1, 64], got %d", lanes);
    c->lanes_override = lanes;
    return PT_OK;
}

int pt_set_mode(pt_context *c, int32_t mode)
{
    if (!c) return PT_ERR_INVALID;
    if (mode != PT_EXACT && mode != PT_FAST) return fail(c, PT_ERR_INVALID, "mode must be PT_EXACT (0) or PT_FAST (1), got %d", mode);
    c->mode = mode;
    return PT_OK;
}

int pt_set_walk_order(pt_context *c, int32_t order)
{
    if (!c) return PT_ERR_INVALID;
    if (order != PT_WALK_REFERENCE && order != PT_WALK_VIEW)
        return fail(c, PT_ERR_INVALID, "walk order must be PT_WALK_REFERENCE (0) or PT_WALK_VIEW (1), got %d", order);
    c->walk_order = order;
    if (c->have_scene) {
        const int rc = thread_nodes(c);
        return rc != PT_OK ? rc : build_primary_lists(c);
    }
    return PT_OK;
}

int pt_set_primary_lists(pt_context *c, int32_t enabled)
{
    if (!c) return PT_ERR_INVALID;
    c->primary_lists = enabled != 0;
    return build_primary_lists(c);
}

int pt_get_primary_lists(const pt_context *c)
{
    if (!c) return PT_ERR_INVALID;
    return c->scene.cand != nullptr ? 1 : 0;
}

int pt_get_walk_order(const pt_context *c)
{
    if (!c) return PT_ERR_INVALID;
    if (!c->have_scene) return c->walk_order;
    return c->threaded_view ? PT_WALK_VIEW : PT_WALK_REFERENCE;
}

int pt_get_lanes_per_pixel(const pt_context *c, int32_t samples, int32_t sample_begin, int32_t sample_end)
{
    if (!c || !c->have_camera || samples < 2 || sample_end <= sample_begin) return PT_ERR_INVALID;
    return pick_lanes(c, sample_end - sample_begin);
}

// pt_render_device / pt_render_device_static: one path-kernel launch whose work-distribution scratch is `sched`
static int render_device_with(pt_context *c, int32_t samples, uint64_t seed, int32_t sample_begin, int32_t sample_end,
                              int32_t accumulate, float *d_radiance, void *stream, int32_t *sched)
{
    int rc = check_ready(c, samples);
    if (rc != PT_OK) return rc;
    if (!d_radiance) return fail(c, PT_ERR_INVALID, "d_radiance is NULL");
    if (sample_begin < 0 || sample_end > samples * samples || sample_begin > sample_end)
        return fail(c, PT_ERR_INVALID, "sample range [%d, %d) not inside [0, %d)", sample_begin, sample_end, samples * samples);
    PT_HIP(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    RenderParams rp;
    rp.seed = seed;
    rp.S = samples;
    rp.s0 = sample_begin;
    rp.s1 = sample_end;
    rp.lanes = pick_lanes(c, sample_end - sample_begin);
    rp.accumulate = accumulate ? 1 : 0;
    rp.inv_s1 = 1.0f / (float)(samples - 1);
    rp.inv_w = 1.0f / (float)c->camera.width;
    rp.inv_h = 1.0f / (float)c->camera.height;
    rp.max_bounces = c->integrator.max_bounces;
    rp.background_kind = c->integrator.background_kind;
    rp.rr_start = c->integrator.rr_start;
    rp.bg_r = c->integrator.background[0];
    rp.bg_g = c->integrator.background[1];
    rp.bg_b = c->integrator.background[2];
    rp.stats = nullptr;
#ifdef PT_STATS
    static unsigned long long *d_stats = nullptr;
    if (!d_stats) (void)hipMalloc((void **)&d_stats, 128);
    (void)hipMemsetAsync(d_stats, 0, 128, st);
    rp.stats = d_stats;
#endif
    // A stream under HIP graph capture records the memset and the kernel launch and nothing else: no events (a
    // captured event cannot be queried or timed afterwards) and no in-flight check (ordering is the graph's).
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &capture) != hipSuccess) { (void)hipGetLastError(); capture = hipStreamCaptureStatusNone; }
    const bool capturing = capture == hipStreamCaptureStatusActive;
    const bool is_static = sched == c->static_sched;
    // One launch at a time per scratch: launches on one stream are ordered; a launch on ANOTHER stream
    // while the previous one is still in flight would clear the tickets under it -- refused, not raced.  The
    // context's own scratch is guarded by the context's record, the library's static array (shared by every
    // context of the GPU) by the array's own record in pt_kernels.hip.
    if (!capturing) {
        if (!is_static && c->own_used && c->own_stream != st && hipEventQuery(c->ev_own_done) == hipErrorNotReady)
            return fail(c, PT_ERR_INVALID, "the previous path-kernel launch of this context is still running on another stream: "
                                           "its launches share one work-distribution scratch (one stream at a time)");
        if (is_static) {
            bool busy = false;
            PT_HIP(c, ptk::static_claim(c->device, st, &busy));
            if (busy)
                return fail(c, PT_ERR_INVALID, "the library's static scratch array on device %d is still in use by a launch on another "
                                               "stream (pt_render_device_static: all contexts of a GPU share it, one stream at a time)", c->device);
        }
        PT_HIP(c, hipEventRecord(c->ev_start, st));
    }
    c->last_sched = sched;
    c->last_stream = st;
    if (c->mode == PT_FAST && !c->scene.ext) PT_HIP(c, ptk_fast::launch_path(c->scene, c->camera, rp, sched, d_radiance, st));
    else PT_HIP(c, ptk::launch_path(c->scene, c->camera, rp, sched, d_radiance, st));
    if (!capturing) {
        PT_HIP(c, hipEventRecord(c->ev_stop, st));
        if (is_static) PT_HIP(c, ptk::static_mark(c->device, st));
        else {
            PT_HIP(c, hipEventRecord(c->ev_own_done, st));
            c->own_stream = st;
            c->own_used = true;
        }
    }
    c->timed = !capturing;
#ifdef PT_STATS
    {
        unsigned long long hs[16];
        (void)hipMemcpyAsync(hs, d_stats, 128, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        const char *nm[4] = {"SPLIT", "LEAF", "SHADE:hit", "SHADE:end"};
        for (int k = 0; k < 4; k++)
            fprintf(stderr, "[pt stats] %-5s steps %llu  lanes/step %.1f\n", nm[k], hs[2 * k], hs[2 * k] ? (double)hs[2 * k + 1] / hs[2 * k] : 0.0);
        const char *tn[6] = {"SPLIT", "LEAF", "SHADE:hit", "SHADE:end", "vote", "tails"};
        double tot = 0;
        for (int k = 0; k < 6; k++) tot += (double)hs[8 + k];
        for (int k = 0; k < 6; k++)
            fprintf(stderr, "[pt stats] wave cycles in %-5s %5.1f %%  (%.0f cycles per step)\n", tn[k], 100.0 * hs[8 + k] / tot, k < 4 && hs[2 * k] ? (double)hs[8 + k] / hs[2 * k] : 0.0);
    }
#endif
    return PT_OK;
}

int pt_render_device(pt_context *c, int32_t samples, uint64_t seed, int32_t sample_begin, int32_t sample_end,
                     int32_t accumulate, float *d_radiance, void *stream)
{
    if (!c) return PT_ERR_INVALID;
    return render_device_with(c, samples, seed, sample_begin, sample_end, accumulate, d_radiance, stream, c->d_sched);
}

int pt_render_device_static(pt_context *c, int32_t samples, uint64_t seed, int32_t sample_begin, int32_t sample_end,
                            int32_t accumulate, float *d_radiance, void *stream)
{
    if (!c) return PT_ERR_INVALID;
    if (!c->static_sched) return fail(c, PT_ERR_HIP, "the library's scratch array was not resolved on device %d", c->device);
    if (c->have_camera && ptk::sched_bytes(c->camera) > c->static_sched_bytes)
        return fail(c, PT_ERR_LIMIT, "pt_render_device_static: a %d x %d frame needs %zu bytes of scratch, the built-in array holds %zu "
                                     "(frames up to %d 8x8 tiles); use pt_render_device",
                    c->camera.width, c->camera.height, ptk::sched_bytes(c->camera), c->static_sched_bytes,
                    (int)((c->static_sched_bytes / sizeof(int32_t)) - 64));
    return render_device_with(c, samples, seed, sample_begin, sample_end, accumulate, d_radiance, stream, c->static_sched);
}

int pt_resolve_device(pt_context *c, int32_t samples, const float *d_radiance, uint8_t *d_rgb8, void *stream)
{
    int rc = check_ready(c, samples);
    if (rc != PT_OK) return rc;
    if (!d_radiance || !d_rgb8) return fail(c, PT_ERR_INVALID, "NULL device buffer");
    PT_HIP(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    PT_HIP(c, ptk::launch_resolve(d_radiance, c->camera.width * c->camera.height, samples, d_rgb8, st));
    return PT_OK;
}

int pt_primary_hits_device(pt_context *c, int32_t samples, int32_t sample_index, int32_t *d_hit_id, float *d_hit_t,
                           void *stream)
{
    int rc = check_ready(c, samples);
    if (rc != PT_OK) return rc;
    if (sample_index < 0 || sample_index >= samples * samples)
        return fail(c, PT_ERR_INVALID, "sample index %d not in [0, %d)", sample_index, samples * samples);
    if (!d_hit_id && !d_hit_t) return fail(c, PT_ERR_INVALID, "no output buffer");
    PT_HIP(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    PT_HIP(c, ptk::launch_primary_hits(c->scene, c->camera, samples, sample_index, d_hit_id, d_hit_t, st));
    return PT_OK;
}

int pt_render(pt_context *c, int32_t samples, uint64_t seed, int32_t sample_begin, int32_t sample_end, uint8_t *rgb8,
              float *radiance, int32_t *hit_id)
{
    int rc = check_ready(c, samples);
    if (rc != PT_OK) return rc;
    PT_HIP(c, hipSetDevice(c->device));
    rc = ensure_frame(c);
    if (rc != PT_OK) return rc;
    const size_t px = (size_t)c->frame_pixels;
    if (rgb8 || radiance) {
        rc = pt_render_device(c, samples, seed, sample_begin, sample_end, 0, c->d_radiance, nullptr);
        if (rc != PT_OK) return rc;
    }
    if (rgb8) {
        rc = pt_resolve_device(c, samples, c->d_radiance, c->d_rgb8, nullptr);
        if (rc != PT_OK) return rc;
        PT_HIP(c, hipMemcpyAsync(rgb8, c->d_rgb8, px * 3, hipMemcpyDeviceToHost, c->stream));
    }
    if (radiance) PT_HIP(c, hipMemcpyAsync(radiance, c->d_radiance, px * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (hit_id) {
        rc = pt_primary_hits_device(c, samples, 0, c->d_hit_id, nullptr, nullptr);
        if (rc != PT_OK) return rc;
        PT_HIP(c, hipMemcpyAsync(hit_id, c->d_hit_id, px * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    }
    PT_HIP(c, hipStreamSynchronize(c->stream));
    if (rgb8 || radiance) return ptapi::check_kernel_error(c);
    return PT_OK;
}

int pt_render_progress(pt_context *c, int32_t samples, uint64_t seed, int32_t passes, pt_progress_fn progress, void *user,
                       uint8_t *rgb8, float *radiance)
{
    int rc = check_ready(c, samples);
    if (rc != PT_OK) return rc;
    const int32_t total = samples * samples;
    if (passes < 1 || passes > total) return fail(c, PT_ERR_INVALID, "passes = %d not in [1, %d]", passes, total);
    PT_HIP(c, hipSetDevice(c->device));
    rc = ensure_frame(c);
    if (rc != PT_OK) return rc;
    const size_t px = (size_t)c->frame_pixels;
    for (int32_t k = 0; k < passes; k++) {
        const int32_t begin = (int32_t)((int64_t)total * k / passes), end = (int32_t)((int64_t)total * (k + 1) / passes);
        rc = pt_render_device(c, samples, seed, begin, end, k > 0, c->d_radiance, nullptr);
        if (rc != PT_OK) return rc;
        PT_HIP(c, hipStreamSynchronize(c->stream));
        rc = ptapi::check_kernel_error(c);
        if (rc != PT_OK) return rc;
        if (progress && progress(user, end, total) != 0)
            return fail(c, PT_ERR_CANCELLED, "cancelled by the progress callback after %d of %d samples", end, total);
    }
    if (rgb8) {
        rc = pt_resolve_device(c, samples, c->d_radiance, c->d_rgb8, nullptr);
        if (rc != PT_OK) return rc;
        PT_HIP(c, hipMemcpyAsync(rgb8, c->d_rgb8, px * 3, hipMemcpyDeviceToHost, c->stream));
    }
    if (radiance) PT_HIP(c, hipMemcpyAsync(radiance, c->d_radiance, px * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    PT_HIP(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

float pt_last_path_kernel_ms(pt_context *c)
{
    if (!c || !c->timed) return -1.0f;
    if (hipSetDevice(c->device) != hipSuccess) return -1.0f;
    if (hipEventSynchronize(c->ev_stop) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, c->ev_start, c->ev_stop) != hipSuccess) return -1.0f;
    // the launch is complete: its error word can be read (see pt_render)
#ifdef PT_PHASES
    {   // measurement build: cycle sums the persistent kernel left in the control words
        unsigned long long w[16] = {0};
        if (c->last_sched && hipMemcpy(w, c->last_sched, sizeof w, hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "PT_PHASES ms %.3f items %d waves %llu draw %llu item %llu wait %llu\n", ms, (int)(w[0] & 0xffffffffu), w[10], w[8], w[9], w[11]);
    }
#endif
    int32_t ctrl[2] = {0, 0};
    if (c->last_sched && hipMemcpy(ctrl, c->last_sched, sizeof ctrl, hipMemcpyDeviceToHost) == hipSuccess && ctrl[1] != 0) {
        (void)hipMemset(c->last_sched, 0, c->last_sched == c->static_sched ? c->static_sched_bytes : c->sched_bytes);
        (void)fail(c, PT_ERR_HIP, ctrl[1] == 2 ? "path kernel: dynamic LDS does not start at address 0"
                                               : "path kernel: a piece gave up waiting for its predecessor");
        return -2.0f;
    }
    return ms;
}

}  // extern "C"
