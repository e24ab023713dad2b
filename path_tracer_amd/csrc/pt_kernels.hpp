// pt_kernels.hpp -- launch interface between the C ABI (pt_api.cpp) and the kernels.
#pragma once
#include "pt_device.hpp"

// ---- LDS layout of a path-kernel workgroup, shared by the kernels (pt_kernels.hip) and the host's node numbering (pt_api.cpp)
#ifndef PT_WAVES_PER_GROUP
#define PT_WAVES_PER_GROUP 16
#endif
#ifndef PT_LDS_N1_OFFSET
#define PT_LDS_N1_OFFSET 32768   // n0[] of the staged nodes from byte 0 of the group's allocation, n1[] from here
#endif
#define PT_PRIMARY_K 8           // primary-ray candidates per 8x8 pixel block (Scene::cand); more: that block walks the tree
// each wave's chain of candidates: per candidate a Split entry holding the box of the Leaf's PARENT and the Leaf's own
// entry -- 2 * PT_PRIMARY_K node entries at the TOP of each half (n0 part below PT_LDS_N1_OFFSET, n1 part
// PT_LDS_N1_OFFSET behind it, like any staged node)
#define PT_CHAIN_ENTRIES (2 * PT_PRIMARY_K)
#define PT_CHAIN_BYTES (16 * PT_CHAIN_ENTRIES * PT_WAVES_PER_GROUP)
#define PT_CHAIN_BASE (PT_LDS_N1_OFFSET - PT_CHAIN_BYTES)
#define PT_MAX_LDS_NODES (PT_CHAIN_BASE / 16)        // 1792 nodes staged at most
#define PT_FIRST_GLOBAL_SLOT (PT_LDS_N1_OFFSET / 16)  // a node that is not staged has a slot from here on: every byte offset
                                                      // below PT_LDS_N1_OFFSET is an LDS address (staged node or chain entry),
                                                      // and the code that ends a walk is never below it (Scene::end_slot)

struct RenderParams {
    uint64_t seed;
    int32_t S;            // side of the sample grid (raytracer.rs:77-81); spp = S*S
    int32_t s0, s1;       // sample-index range [s0, s1) rendered by this launch
    int32_t lanes;        // lanes per pixel: power of two in [1, 64]
    int32_t accumulate;   // continue the sum already in the radiance buffer
    float inv_s1, inv_w, inv_h;   // 1/(S-1), 1/W, 1/H: correctly rounded fp32 reciprocals (host), for div_by
    // extension scenes only (pt_set_integrator; defaults = the reference's constants)
    int32_t max_bounces, background_kind, rr_start;
    float bg_r, bg_g, bg_b;
    unsigned long long *stats;   // PT_STATS builds only: {steps, active lanes} x {SPLIT, LEAF, SHADE:hit, SHADE:end}, then 6 cycle buckets
};

// Work distribution of the path kernel: one wave per (piece, tile) item, items handed out
// piece-major through one ticket counter in the order the workgroups start.  A piece is a contiguous run of `piece_len` sample indices;
// piece k of a tile continues the per-pixel sums piece k-1 left in the radiance buffer
// (same sequential fp32 order as one long run) and therefore waits for done[tile] >= k.
struct SchedParams {
    int32_t n_tiles, n_pieces, piece_len;
    int32_t *ctrl;        // [0] next ticket, [1] error flag (a dependency wait gave up)
    int32_t *done;        // per tile: pieces completed
};

namespace ptk {
int tiles_for(const ptd::Camera &cam, int lanes);
// counts, into *d_count, the fp32 bit patterns for which short form `fn` differs from the IEEE operation
hipError_t launch_unary_selftest(int fn, unsigned long long *d_count, hipStream_t stream);
// counts the operand pairs (2^log2_pairs of them) on which the reciprocal-based division differs from x / a
hipError_t launch_division_selftest(int log2_pairs, unsigned long long *d_count, hipStream_t stream);
// `groups` workgroups of 4 waves, each leaving {shader cycles, 100 MHz ticks} of a multiply chain of `iters` rounds in d_stamps
hipError_t launch_clock_probe(unsigned long long *d_stamps, float *d_sink, int groups, int iters, hipStream_t stream);
hipError_t launch_grid_selftest(int width, int height, int S, float inv_s1, float inv_w, float inv_h, unsigned long long *d_count,
                                hipStream_t stream);
hipError_t launch_path(const ptd::Scene &sc, const ptd::Camera &cam, const RenderParams &rp, int32_t *d_sched,
                       float *d_radiance, hipStream_t stream);
size_t sched_bytes(const ptd::Camera &cam);
// does launch_path render this scene with the persistent kernel (which stamps its clock) or with the small-tree one?
bool launch_is_persistent(const ptd::Scene &sc);
// address and size of the library's file-scope scratch array on the current device (pt_render_device_static)
hipError_t static_sched(int device, int32_t **out, size_t *bytes);
// the array's in-flight record, shared by every context of the device: *busy = its previous launch went to another
// stream and has not completed; static_mark records the launch just enqueued on `stream` as the most recent one
hipError_t static_claim(int device, hipStream_t stream, bool *busy);
hipError_t static_mark(int device, hipStream_t stream);
hipError_t launch_primary_hits(const ptd::Scene &sc, const ptd::Camera &cam, int S, int sample, int32_t *d_hit_id,
                               float *d_hit_t, hipStream_t stream);
// fills lists[blocks_x * blocks_y][PT_PRIMARY_K] for the camera (Scene::cand); leaf_codes[rank] = code of the primitive's Leaf
hipError_t launch_primary_lists(const ptd::Scene &sc, const ptd::Camera &cam, const int32_t *d_leaf_codes, int n_leaves,
                                int32_t *d_lists, hipStream_t stream);
hipError_t launch_resolve(const float *d_radiance, int n_pixels, int S, uint8_t *d_rgb8, hipStream_t stream);
}  // namespace ptk

// the FAST build of the path kernel (pt_kernels_fast.hip; reference scenes only)
namespace ptk_fast {
hipError_t launch_path(const ptd::Scene &sc, const ptd::Camera &cam, const RenderParams &rp, int32_t *d_sched,
                       float *d_radiance, hipStream_t stream);
}  // namespace ptk_fast
