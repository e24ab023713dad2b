// pt_kernels.hip -- gfx950 kernels of the radiance hot path and their launchers.
//
//   K1 primary_hit_kernel : camera.ray + Scene::trace for one sample index (AOV)
//   K2 path_kernel_persistent (path_kernel for trees of a few nodes): trace_pixel + trace_ray and
//                           everything below (src/raytracer.rs:46-96), one lane per (pixel, sample chunk)
//   K3 resolve_kernel     : sqrt(sum / S^2) * 255 -> u8 (src/raytracer.rs:92-93,119-121)
//
// Compiled with -ffp-contract=off (see pt_device.hpp).
#include <mutex>

#include "pt_kernels.hpp"

using namespace ptd;

// This file is compiled twice: as itself (EXACT: every kernel of the library) and, through
// pt_kernels_fast.hip, with PT_FAST_MODE defined and contraction on -- then only the path kernel of the
// reference's scenes is built, behind ptk_fast::launch_path (pt_set_mode(PT_FAST)).
#ifdef PT_FAST_MODE
#ifndef PT_FAST_BOX
#define PT_FAST_BOX 1                 // 1: the slab test with one fma per plane (split_run_regular_fast)
#endif
#ifndef PT_FAST_BOX_LIMIT
#define PT_FAST_BOX_LIMIT 4096.0f     // largest |o * inv| a lane may have for its wave to use it: error <= 2^-12 in t
#endif
#define PTK ptk_fast
#define path_kernel path_kernel_fast   // its own name in profiles
#define path_kernel_persistent path_kernel_persistent_fast
#else
#define PTK ptk
#endif

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
//   SHADE  the walk has ended: Sphere::record + Material::scatter on a hit (sphere.rs:37-42,
//          material/*.rs); sky, accumulate, next sample's camera ray when the path ends
//          (raytracer.rs:67-70,88,77-85)
// A lane whose path or walk ends never idles until the longest one of the wave
// ends, and box tests / sphere tests / shading are not serialised against each
// other inside one iteration.  Each lane still executes exactly the reference's
// operation sequence for its pixel, so results do not depend on the vote.
#ifndef PT_SPLIT_KEEP
#define PT_SPLIT_KEEP 3   // a SPLIT run goes on while at least KEEP/8 of the lanes it started with are still at a Split
#endif
#ifndef PT_PIECE_MIN
#define PT_PIECE_MIN 64       // fewest sample indices in a piece (each piece ends with a drain of the wave, and a tile's
                              // pieces run one after another): 32 -> 64 is -12 % on cfg 2 (256 spp: 4 pieces instead of 8),
                              // -1.8 % at 529 spp (one GPU's share of the headline frame on 8), nothing at 1024 and 4096 spp
#endif
#ifndef PT_SPLIT_UNROLL
#define PT_SPLIT_UNROLL 1   // two node visits per trip of the SPLIT run (split_run_regular)
#endif
#ifndef PT_ITEMS_PER_SLOT
#define PT_ITEMS_PER_SLOT 64  // (piece, tile) items wanted per wave slot of the chip
#endif
#ifndef PT_LEAF_KEEP
#define PT_LEAF_KEEP 2    // the same for a LEAF run (a Leaf's next node is often its sibling Leaf): -4.5 %
#endif
// vote weights: a SHADE step is the most expensive one, so it waits for more lanes
#ifndef PT_FAST_DIV
#define PT_FAST_DIV 1
#endif
#ifndef PT_W_SPLIT
#define PT_W_SPLIT 3
#endif
#ifndef PT_W_LEAF
#define PT_W_LEAF 5
#endif
#ifndef PT_W_SHADE
#define PT_W_SHADE 1
#endif
// the same for the extension's scenes (a rect test of an instance costs more than a sphere test; walks are short)
#ifndef PT_W_SPLIT_EXT
#define PT_W_SPLIT_EXT 3
#endif
#ifndef PT_W_LEAF_EXT
#define PT_W_LEAF_EXT 5
#endif
#ifndef PT_W_SHADE_EXT
#define PT_W_SHADE_EXT 1
#endif
#ifndef PT_LEAF_KEEP_EXT
#define PT_LEAF_KEEP_EXT 4
#endif
#ifndef PT_SPLIT_KEEP_EXT
#define PT_SPLIT_KEEP_EXT PT_SPLIT_KEEP
#endif


// The launch is PERSISTENT: as many workgroups as the chip holds at once (occupancy x CUs), each of 16 waves
// that share one LDS copy of the top of the tree (breadth-first, lds_nodes nodes) and are otherwise independent:
// every wave draws (piece, tile) items from one counter until none are left (path_kernel_persistent below).
// Against one workgroup of 4 waves per 4 items (the round-1 launch: 516 ms for the headline frame) no wave slot
// waits for the slowest wave of its group and the tree is staged once per slot instead of once per item: 492 ms
// at any group size (4, 8, 16 waves).
constexpr int WAVES_PER_GROUP = PT_WAVES_PER_GROUP;   // pt_kernels.hpp
// BOTH halves of a staged node sit in LDS -- n0[] from byte 0, n1[] from byte N1OFF of the group's allocation (a
// template parameter: PT_LDS_N1_OFFSET for the persistent launch, PT_SMALL_N1_OFFSET for the small-tree one) -- so
// a visit of a staged node makes no vector-memory access at all: 482 ms.  (Without the persistent launch it did
// not pay: the 8-wave groups it needs to keep 8 waves per SIMD lost more to group tails than the load cost -- 561
// ms against 521 ms for groups of 4 with n1[] through the L1.)  A tree of more than PT_LDS_N1_OFFSET / 16 nodes
// keeps that many, its top breadth-first, in LDS and reads both halves of the others through the L1: a 16-byte
// gather costs the CU 10 + 0.85 cycles per ENABLED lane (tools/ubench/vmem_sparse.hip), so the lanes standing on
// staged nodes -- 85 % of the visits of cfg 5's 21,385-node tree -- are left out of it.
#ifndef PT_SMALL_TREE
#define PT_SMALL_TREE 32   // nodes; below it the launch without the item loop (launch_path)
#endif
#define PT_SMALL_N1_OFFSET (16 * PT_SMALL_TREE)
#define PT_STR2(x) #x
#define PT_STR(x) PT_STR2(x)
#ifndef PT_OBJ_CACHE
#define PT_OBJ_CACHE 1    // the extension's kernel keeps a walk's object-space rays in LDS (pt_device.hpp obj_cache_fill)
#endif
#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 8   // waves per SIMD the register allocator must leave room for (64 VGPRs)
#endif

// SPLIT run for REGULAR rays (pt_device.hpp split_step<true>, one AABB::intersect per iteration),
// hand-scheduled.  Entered by the lanes standing on a Split; a lane leaves (exec shrinks) when
// its code stops being a Split, and the run ends for all once fewer than `keep` remain; exec is
// restored on exit.  21 (ALL_IN_LDS) / 22 VALU + 3 / 8 SALU + one branch per node, against the
// compiler's 23 + 7 for the same C++ loop: the scalar unit is shared by the CU's four SIMDs, so
// loop control costs as much as arithmetic (tools/ubench/salu_rate.hip).  -5 %.
//   Both halves of a staged node come from LDS (n0 at the node's code, n1 N1OFF bytes behind it); when the
//   tree is larger than the staged prefix, the lanes outside it read both halves through the L1, under their
//   own exec mask, and those loads are skipped when no lane needs them.
// The arithmetic is split_step<true>'s, instruction for instruction; v48-v57 are scratch.
template <bool ALL_IN_LDS, int N1OFF>
__device__ __forceinline__ int32_t split_run_regular(int32_t code, const f3 o, const f3 inv, const float closest, const Scene &sc,
                                                     const int n_split_end, const int keep)
{
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const uint64_t b1 = (uint64_t)(uintptr_t)sc.n1, b0 = (uint64_t)(uintptr_t)sc.n0;
    const v4u rs1 = {(uint32_t)b1, (uint32_t)(b1 >> 32) & 0xffffu, (uint32_t)(sc.n_nodes * 16), 0x00020000u};
    const v4u rs0 = {(uint32_t)b0, (uint32_t)(b0 >> 32) & 0xffffu, (uint32_t)(sc.n_nodes * 16), 0x00020000u};
    uint64_t save, cur; int cnt;
// Scratch registers of the run: A = n0 (v48-51), B = n1 (v52-55), E / F = the tmin / tmax chains (v56, v57).
// Each pair (n0.k, n1.k) shares a VGPR bank (register number mod 4); a layout in which no instruction reads two
// scratch registers of one bank (A = v46-49, B = v52-55, E = v51, F = v57, chains y, x, z) measured the same
// (520.6 vs 521.0 ms): operand banks do not limit this loop.  What an instruction MORE in this loop cost, each
// measured by adding one to the 522 ms frame of the launch that read n1 through the L1: a vector-memory load of
// any width +110 ms (the same 16 bytes as a 12- and a 4-byte load: 632 ms); an LDS read of 4, 8 or 16 bytes:
// nothing (522-526 ms).
#define RA0 "v48"
#define RA1 "v49"
#define RA2 "v50"
#define RA3 "v51"
#define RA "v[48:51]"
#define RB0 "v52"
#define RB1 "v53"
#define RB2 "v54"
#define RB3 "v55"
#define RB "v[52:55]"
#define RE "v56"
#define RF "v57"
#define PT_RUN_CLOBBERS "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57"
#define PT_RUN_CHAINS \
    "v_med3_f32 " RE ", " RA0 ", " RB0 ", %[tmin]\n v_med3_f32 " RF ", %[t], " RA0 ", " RB0 "\n" \
    "v_med3_f32 " RE ", " RE ", " RA1 ", " RB1 "\n v_med3_f32 " RF ", " RF ", " RA1 ", " RB1 "\n" \
    "v_med3_f32 " RE ", " RE ", " RA2 ", " RB2 "\n v_med3_f32 " RF ", " RF ", " RA2 ", " RB2 "\n"
#define PT_RUN_TEST \
    "v_sub_f32 " RA0 ", " RA0 ", %[ox]\n v_sub_f32 " RA1 ", " RA1 ", %[oy]\n v_sub_f32 " RA2 ", " RA2 ", %[oz]\n" \
    "v_mul_f32 " RA0 ", %[ix], " RA0 "\n v_mul_f32 " RA1 ", %[iy], " RA1 "\n v_mul_f32 " RA2 ", %[iz], " RA2 "\n" \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n" \
    "v_sub_f32 " RB0 ", " RB0 ", %[ox]\n v_sub_f32 " RB1 ", " RB1 ", %[oy]\n v_sub_f32 " RB2 ", " RB2 ", %[oz]\n" \
    "v_mul_f32 " RB0 ", %[ix], " RB0 "\n v_mul_f32 " RB1 ", %[iy], " RB1 "\n v_mul_f32 " RB2 ", %[iz], " RB2 "\n" \
    PT_RUN_CHAINS \
    "v_cmp_lt_f32 vcc, " RE ", " RF "\n s_nop 1\n" \
    "v_cndmask_b32 %[code], " RA3 ", " RB3 ", vcc\n" \
    PT_RUN_TAIL
#define PT_RUN_TAIL_COUNT \
    "v_cmp_gt_u32 vcc, %[nspl], %[code]\n" \
    "s_bcnt1_i32_b64 %[cnt], vcc\n s_cmp_ge_i32 %[cnt], %[keep]\n s_cselect_b64 exec, vcc, 0\n"
#define PT_RUN_TAIL PT_RUN_TAIL_COUNT
    if (ALL_IN_LDS) {
        asm volatile("s_mov_b64 %[save], exec\n"
                     ".LPTRUN%=:\n"
#if PT_SPLIT_UNROLL
// Two visits per trip: the first leaves through exec alone (v_cmpx: the lanes still on a Split), only the second
// counts them against `keep` -- three scalar instructions fewer per two visits, and a run may go one visit past the
// count that would have ended it, with the lanes that are still in it: -0.4 to -1.0 % in four A/B runs (three or four
// visits per trip: no more).
#undef PT_RUN_TAIL
#define PT_RUN_TAIL "v_cmpx_gt_u32 vcc, %[nspl], %[code]\n"
                     "ds_read_b128 " RA ", %[code]\n"
                     "ds_read_b128 " RB ", %[code] offset:%[n1off]\n"
                     "s_waitcnt lgkmcnt(1)\n"
                     PT_RUN_TEST
                     "s_cbranch_execz .LPTRUNX%=\n"
#undef PT_RUN_TAIL
#define PT_RUN_TAIL PT_RUN_TAIL_COUNT
#endif
                     "ds_read_b128 " RA ", %[code]\n"
                     "ds_read_b128 " RB ", %[code] offset:%[n1off]\n"
                     "s_waitcnt lgkmcnt(1)\n"
                     PT_RUN_TEST
                     "s_cbranch_execnz .LPTRUN%=\n"
                     ".LPTRUNX%=:\n"
                     "s_mov_b64 exec, %[save]\n"
                     : [code] "+v"(code), [save] "=&s"(save), [cnt] "=&s"(cnt)
                     : [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z), [ix] "v"(inv.x), [iy] "v"(inv.y), [iz] "v"(inv.z), [t] "v"(closest),
                       [tmin] "s"(T_MIN), [nspl] "s"(n_split_end), [keep] "s"(keep), [n1off] "i"(N1OFF)
                     : "vcc", "scc", "memory", PT_RUN_CLOBBERS);
    } else {
        asm volatile("s_mov_b64 %[save], exec\n"
                     ".LPTRUN%=:\n"
                     "v_cmp_gt_u32 vcc, %[lim], %[code]\n"
                     "s_mov_b64 %[cur], exec\n"
                     "s_andn2_b64 exec, %[cur], vcc\n"
                     "s_cbranch_execz .LPTNODEEP%=\n"
                     "buffer_load_dwordx4 " RB ", %[code], %[rs1], 0 offen\n"
                     "buffer_load_dwordx4 " RA ", %[code], %[rs0], 0 offen\n"
                     ".LPTNODEEP%=:\n"
                     "s_and_b64 exec, %[cur], vcc\n"
                     "ds_read_b128 " RA ", %[code]\n"
                     "ds_read_b128 " RB ", %[code] offset:%[n1off]\n"
                     "s_mov_b64 exec, %[cur]\n"
                     "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
                     PT_RUN_TEST
                     "s_cbranch_execnz .LPTRUN%=\n"
                     "s_mov_b64 exec, %[save]\n"
                     : [code] "+v"(code), [save] "=&s"(save), [cur] "=&s"(cur), [cnt] "=&s"(cnt)
                     : [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z), [ix] "v"(inv.x), [iy] "v"(inv.y), [iz] "v"(inv.z), [t] "v"(closest),
                       [rs0] "s"(rs0), [rs1] "s"(rs1), [tmin] "s"(T_MIN), [nspl] "s"(n_split_end), [keep] "s"(keep),
                       [lim] "s"(N1OFF), [n1off] "i"(N1OFF)
                     : "vcc", "scc", "memory", PT_RUN_CLOBBERS);
    }
#undef PT_RUN_TEST
    (void)rs0; (void)rs1; (void)cur;
    return code;
}

#ifdef PT_FAST_MODE
// The SPLIT run of the FAST build: one v_fma_f32 per plane (pt_device.hpp split_step_fast): 15 VALU per node.
template <bool ALL_IN_LDS, int N1OFF>
__device__ __forceinline__ int32_t split_run_regular_fast(int32_t code, const f3 inv, const f3 noi, const float closest, const Scene &sc,
                                                          const int n_split_end, const int keep)
{
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const uint64_t b1 = (uint64_t)(uintptr_t)sc.n1, b0 = (uint64_t)(uintptr_t)sc.n0;
    const v4u rs1 = {(uint32_t)b1, (uint32_t)(b1 >> 32) & 0xffffu, (uint32_t)(sc.n_nodes * 16), 0x00020000u};
    const v4u rs0 = {(uint32_t)b0, (uint32_t)(b0 >> 32) & 0xffffu, (uint32_t)(sc.n_nodes * 16), 0x00020000u};
    uint64_t save, cur; int cnt;
#define PT_RUN_TEST_FAST \
    "v_fma_f32 v48, v48, %[ix], %[nx]\n v_fma_f32 v49, v49, %[iy], %[ny]\n v_fma_f32 v50, v50, %[iz], %[nz]\n" \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n" \
    "v_fma_f32 v52, v52, %[ix], %[nx]\n v_fma_f32 v53, v53, %[iy], %[ny]\n v_fma_f32 v54, v54, %[iz], %[nz]\n" \
    "v_med3_f32 v56, v48, v52, %[tmin]\n v_med3_f32 v57, %[t], v48, v52\n" \
    "v_med3_f32 v56, v56, v49, v53\n v_med3_f32 v57, v57, v49, v53\n" \
    "v_med3_f32 v56, v56, v50, v54\n v_med3_f32 v57, v57, v50, v54\n" \
    "v_cmp_lt_f32 vcc, v56, v57\n s_nop 1\n" \
    "v_cndmask_b32 %[code], v51, v55, vcc\n" \
    PT_RUN_TAIL
    if (ALL_IN_LDS) {
        asm volatile("s_mov_b64 %[save], exec\n"
                     ".LPTRUNF%=:\n"
#if PT_SPLIT_UNROLL   // two visits per trip, as in split_run_regular
#undef PT_RUN_TAIL
#define PT_RUN_TAIL "v_cmpx_gt_u32 vcc, %[nspl], %[code]\n"
                     "ds_read_b128 v[48:51], %[code]\n"
                     "ds_read_b128 v[52:55], %[code] offset:%[n1off]\n"
                     "s_waitcnt lgkmcnt(1)\n"
                     PT_RUN_TEST_FAST
                     "s_cbranch_execz .LPTRUNFX%=\n"
#undef PT_RUN_TAIL
#define PT_RUN_TAIL PT_RUN_TAIL_COUNT
#endif
                     "ds_read_b128 v[48:51], %[code]\n"
                     "ds_read_b128 v[52:55], %[code] offset:%[n1off]\n"
                     "s_waitcnt lgkmcnt(1)\n"
                     PT_RUN_TEST_FAST
                     "s_cbranch_execnz .LPTRUNF%=\n"
                     ".LPTRUNFX%=:\n"
                     "s_mov_b64 exec, %[save]\n"
                     : [code] "+v"(code), [save] "=&s"(save), [cnt] "=&s"(cnt)
                     : [nx] "v"(noi.x), [ny] "v"(noi.y), [nz] "v"(noi.z), [ix] "v"(inv.x), [iy] "v"(inv.y), [iz] "v"(inv.z), [t] "v"(closest),
                       [tmin] "s"(T_MIN), [nspl] "s"(n_split_end), [keep] "s"(keep), [n1off] "i"(N1OFF)
                     : "vcc", "scc", "memory", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57");
    } else {
        asm volatile("s_mov_b64 %[save], exec\n"
                     ".LPTRUNF%=:\n"
                     "v_cmp_gt_u32 vcc, %[lim], %[code]\n"
                     "s_mov_b64 %[cur], exec\n"
                     "s_andn2_b64 exec, %[cur], vcc\n"
                     "s_cbranch_execz .LPTNODEEPF%=\n"
                     "buffer_load_dwordx4 v[52:55], %[code], %[rs1], 0 offen\n"
                     "buffer_load_dwordx4 v[48:51], %[code], %[rs0], 0 offen\n"
                     ".LPTNODEEPF%=:\n"
                     "s_and_b64 exec, %[cur], vcc\n"
                     "ds_read_b128 v[48:51], %[code]\n"
                     "ds_read_b128 v[52:55], %[code] offset:%[n1off]\n"
                     "s_mov_b64 exec, %[cur]\n"
                     "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
                     PT_RUN_TEST_FAST
                     "s_cbranch_execnz .LPTRUNF%=\n"
                     "s_mov_b64 exec, %[save]\n"
                     : [code] "+v"(code), [save] "=&s"(save), [cur] "=&s"(cur), [cnt] "=&s"(cnt)
                     : [nx] "v"(noi.x), [ny] "v"(noi.y), [nz] "v"(noi.z), [ix] "v"(inv.x), [iy] "v"(inv.y), [iz] "v"(inv.z), [t] "v"(closest),
                       [rs0] "s"(rs0), [rs1] "s"(rs1), [tmin] "s"(T_MIN), [nspl] "s"(n_split_end), [keep] "s"(keep),
                       [lim] "s"(N1OFF), [n1off] "i"(N1OFF)
                     : "vcc", "scc", "memory", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57");
    }
#undef PT_RUN_TEST_FAST
    (void)rs0; (void)rs1; (void)cur;
    return code;
}
#endif

// One (piece, tile) item, rendered by one wave.  The kernels below hand the items out: path_kernel one per wave
// of a workgroup that then ends, path_kernel_persistent in a loop.
template <bool ALL_IN_LDS, bool EXT, int N1OFF>
__device__ __forceinline__ void render_item(const Scene &sc, const Camera &cam, const RenderParams &rp, const SchedParams &sched,
                                            float *radiance, const int item, const uint32_t xf_lds)
{
    extern __shared__ float4 lds_n0[];
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const v4f lds_f4;
    auto lds_read = [](uint32_t addr) -> float4 { const v4f t = *(lds_f4 *)(uintptr_t)addr; return make_float4(t.x, t.y, t.z, t.w); };
    typedef __attribute__((address_space(3))) v4f lds_f4_rw;
    auto lds_write = [](uint32_t addr, float4 v) { v4f t = {v.x, v.y, v.z, v.w}; *(lds_f4_rw *)(uintptr_t)addr = t; };
    (void)xf_lds;
    const int lane = threadIdx.x & 63;
    const int L = rp.lanes;
    const int P = 64 / L;
    int tw, th;
    tile_shape(L, tw, th);
    const int slot = lane % P, chunk = lane / P;
    const int tiles_x = (cam.width + tw - 1) / tw;
    const int S = rp.S;
    const int n_nodes = sc.end_slot << PT_CODE_SHIFT;  // the code that ends a walk; every Split code -- a node's or a chain entry's -- is below it
    const __amdgpu_buffer_rsrc_t n1_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)sc.n1, 0, sc.n_nodes * 16, 0x00020000);

  // A wave renders ONE item: a tile's pixels x a contiguous piece of the sample indices.
  // Fine-grained items keep every SIMD's wave slots full until the very end of the frame (a
  // whole tile x all samples per wave left the chip at 4.75 of 8 waves per SIMD on average:
  // profiles/r01/v2); a loop over items inside the wave cost 4-20 % in code quality.
  {
    const int piece = item / sched.n_tiles, tile = item - piece * sched.n_tiles;
    const int ps0 = rp.s0 + piece * sched.piece_len;
    const int ps1 = ps0 + sched.piece_len < rp.s1 ? ps0 + sched.piece_len : rp.s1;
    const int px = (tile % tiles_x) * tw + slot % tw;
    const int py = (tile / tiles_x) * th + slot / tw;
    const bool valid = px < cam.width && py < cam.height;
    const uint32_t pixel = (uint32_t)(py * cam.width + px);

    const int per = (ps1 - ps0 + L - 1) / L;
    int s = ps0 + chunk * per;
    int s_end = s + per < ps1 ? s + per : ps1;
    if (!valid) s_end = s;
    // grid coordinates of sample s, kept incrementally: (sx << 16) | sy, s = sx * S + sy (sx outer)
    uint32_t sxy = ((uint32_t)(s / S) << 16) | (uint32_t)(s % S);
    const bool resume = rp.accumulate || piece > 0;

#ifdef PT_PHASES
    const unsigned long long tw0 = clock64();
#endif
    if (piece > 0) {
        // the previous piece of this tile was drawn earlier, by a wave that is running: wait for
        // its sums (agent-scope acquire: cdna_hip_programming.md guideline 16); bounded spin
        int spins = 0;
        while (__hip_atomic_load(&sched.done[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < piece) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > (1 << 22)) { sched.ctrl[1] = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

#ifdef PT_PHASES
    if (lane == 0) atomicAdd((unsigned long long *)&sched.ctrl[22], (unsigned long long)(clock64() - tw0));
#endif
    f3 sum = fill(0.0f);
    if (L == 1 && resume && valid)
        sum = mk(radiance[3 * pixel], radiance[3 * pixel + 1], radiance[3 * pixel + 2]);

    // Per-lane state.  `code` is the whole control state of the lane: a node code while
    // the walk stands on a node (Split: [0, n_nodes), Leaf: sign bit set), n_nodes when
    // the walk has just ended, and three values above it for the other states.
    const int32_t C_DONE = n_nodes + 1;
    Ray ray; ray.o = fill(0.0f); ray.d = fill(0.0f);
    f3 inv = fill(0.0f);
#ifdef PT_FAST_MODE
    f3 noi = fill(0.0f);          // -(o * inv), once per walk (split_step_fast)
    bool fma_ok = true;
#endif
    float a = 0.0f, ra = 0.0f;
    Hit h; h.t = T_MAX; h.prim = -1;
    int32_t code = C_DONE;
    f3 att = fill(1.0f);
    f3 rad = fill(0.0f);          // EXT: radiance the path has collected from lights so far
    int bounce = 0;
    uint64_t rng = 0;
    bool regular = true;

    // both halves of a staged node from LDS, of any other node through the L1 (one 32-bit byte offset serves all)
    // SPLIT_ONLY: the caller knows c is a Split code (no Leaf flag to mask)
    auto fetch = [&](int32_t c, float4 &n0, float4 &n1, const bool SPLIT_ONLY = false) {
        const int off = SPLIT_ONLY ? (c << (4 - PT_CODE_SHIFT)) : code_offset(c);
        if (ALL_IN_LDS || off < N1OFF) {   // every offset below N1OFF is an LDS address: a staged node or an entry of the wave's chain
            n0 = lds_read((uint32_t)off);
            n1 = lds_read((uint32_t)off + (uint32_t)N1OFF);
            if (!ALL_IN_LDS) asm volatile("" : "+v"(n0.x), "+v"(n1.x));
        } else {
            const auto raw = __builtin_amdgcn_raw_buffer_load_b128(n1_rsrc, off, 0, 0);
            n1 = make_float4(__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]), __uint_as_float(raw[3]));
            n0 = *(const float4 *)((const char *)sc.n0 + off);
        }
    };
    auto lanes = [&](bool p) -> int { return __builtin_amdgcn_readfirstlane((int)__popcll(__ballot(p))); };

    // Where a PRIMARY ray's walk starts.  A camera ray through this wave's pixels can only hit the few spheres
    // whose (slightly inflated) balls meet the pixel block's frustum: primary_list_kernel lists their Leaves per 8x8
    // pixel block (2.2 on average on the bench scenes; Scene::cand).  The wave builds a chain of its own in LDS out of
    // them -- per candidate a Split entry carrying the box of the Leaf's PARENT (box passed: the Leaf's entry; missed:
    // the next candidate) and the Leaf's entry (then the next candidate; after the last one the end code) -- and its
    // regular primary rays walk that chain instead of the tree: per candidate the LAST box test the reference's walk
    // makes before it reaches the Leaf, then the same Sphere::trace, in the same order of ranks.
    //   Why the parent's box, and why it is enough: where a sphere TOUCHES its boxes and the ray grazes it there, the
    // slab test is exact and Sphere::trace is not (its discriminant is a difference of two large numbers), so the sphere
    // test "hits" for rays that pass just outside -- rays the reference never gets to test, because its walk stops at
    // the box.  main.rs's eye sits at y = 2, the height of the big spheres' tops: the full cfg 5 frame has a row of
    // horizontal rays a few 1e-6 above those poles, and differed in 11 pixels while the chain held Leaves only.  The
    // boxes filter what the sphere test sees, and the filter has to stay.  But the boxes on the way are nested (a
    // Split's box is the min / max of its children's: exact in fp32) and the slab test is monotone in the box for a ray
    // without NaNs ((b - o) * inv and min / max are monotone under rounding): a ray that passes the parent's test
    // passes every ancestor's.  So "reaches the Leaf" is decided by the parent's box alone; what differs from the
    // reference's walk is only the closest hit a box is tested against at that moment, as with any other visiting
    // order.
    // Rays that are not REGULAR (a NaN or an infinity in the slab test breaks the monotony) walk the tree, and so does
    // a block with more than PT_PRIMARY_K candidates.  The host builds lists only for trees whose boxes are nested.
    int32_t primary_code = sc.root_code;
    if (N1OFF == PT_LDS_N1_OFFSET && !EXT && sc.cand) {
        const int ctile = ((tile / tiles_x) * th >> 3) * sc.cand_tiles_x + ((tile % tiles_x) * tw >> 3);
        const int32_t mine = lane < PT_PRIMARY_K ? sc.cand[(size_t)ctile * PT_PRIMARY_K + lane] : 0;
        const int32_t first = __builtin_amdgcn_readfirstlane(mine);
        if (first != 1) {                                                        // 1: too many candidates, walk the tree
            const int count = lanes(mine != 0);
            // entries 2k (the Split) and 2k + 1 (the Leaf) of this wave's chain, as byte offsets = node codes
            const uint32_t chain = (uint32_t)(PT_CHAIN_BASE + (threadIdx.x >> 6) * (16 * PT_CHAIN_ENTRIES));
            if (mine != 0) {
                const uint32_t split_at = chain + 32u * (uint32_t)lane, leaf_at = split_at + 16u;
                const int32_t after = lane + 1 < count ? (int32_t)(split_at + 32u) : n_nodes;   // the next candidate's Split, or the end
                float4 n0, n1;
                fetch(mine, n0, n1);                                             // the Leaf: (c, r^2), (-, -, rank, next)
                const int32_t parent = sc.parent_codes[__float_as_int(n1.z)];
                n1.w = __int_as_float(after);
                lds_write(leaf_at, n0);
                lds_write(leaf_at + (uint32_t)N1OFF, n1);
                fetch(parent, n0, n1);                                           // its parent: (bmin, skip), (bmax, next)
                n0.w = __int_as_float(after);                                    // box missed: on to the next candidate
                n1.w = __int_as_float((int32_t)(leaf_at | 0x80000000u));         // box passed: this Leaf
                lds_write(split_at, n0);
                lds_write(split_at + (uint32_t)N1OFF, n1);
            }
            primary_code = count > 0 ? (int32_t)chain : n_nodes;                 // no candidate: the ray misses everything
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }

    // SPLIT and LEAF steps go straight back to the vote; SHADE falls through to
    // the shared tails: new_path -> the next sample's camera ray (or DONE),
    // new_walk -> Scene::trace prologue (scene/mod.rs:24-34), standing on the root.
#ifdef PT_STATS
    unsigned st_n[4] = {0, 0, 0, 0}, st_l[4] = {0, 0, 0, 0};
    unsigned long long st_c[6] = {0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
    // The counters live in LANE 0's registers and are updated with v_readlane / v_writelane, which ignore
    // exec: a plain `st_n[k] += 1` inside a step only counts the steps lane 0 itself takes part in (and then
    // averages the lanes per step over those, a size-biased sample).
    auto st_add = [](unsigned v, unsigned n) -> unsigned {
        const unsigned sum = (unsigned)__builtin_amdgcn_readlane((int)v, 0) + (unsigned)__builtin_amdgcn_readfirstlane((int)n);
        asm volatile("s_nop 4\n v_writelane_b32 %0, %1, 0\n s_nop 1" : "+v"(v) : "s"(sum));
        return v;
    };
#define PT_STAT(k, n) do { st_n[k] = st_add(st_n[k], 1u); st_l[k] = st_add(st_l[k], (unsigned)(n)); } while (0)
#define PT_TIME(k) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_c[k] += t_ - st_t; st_t = t_; } while (0)
#else
#define PT_STAT(k, n) do { } while (0)
#define PT_TIME(k) do { } while (0)
#endif
    const uint32_t obj_at = (uint32_t)PT_OBJ_BASE + 8u * threadIdx.x;           // this lane's column of the object-ray cache (EXT)
    bool new_path = true, new_walk = false;
    for (;;) {
        if (new_path) {
            if (s >= s_end) code = C_DONE;
            else {
                ray = sample_ray_grid(cam, S, rp.inv_s1, rp.inv_w, rp.inv_h, px, py, (int)(sxy >> 16), (int)(sxy & 0xffffu));   // raytracer.rs:77-85
                att = fill(1.0f);                                                // raytracer.rs:50
                if (EXT) rad = fill(0.0f);
                bounce = 0;
                rng = path_rng(rp.seed, pixel, (uint32_t)s);
                new_walk = true;
            }
        }
        if (new_walk) {
            inv = mk(p_rcp(ray.d.x), p_rcp(ray.d.y), p_rcp(ray.d.z));            // scene/mod.rs:31: (1,1,1) / dir
            a = dot(ray.d, ray.d);                                               // sphere.rs:49
            ra = walk_rcp(a);                                                    // read by REGULAR rays only
#ifdef PT_FAST_MODE
            // t = b * inv + noi carries the rounding error of o * inv, about 2^-24 |o * inv|: harmless unless a
            // direction component is nearly zero.  A lane whose |o * inv| exceeds 2^PT_FAST_BOX_LOG2 keeps its
            // wave on the exact (b - o) * inv test for this walk.
            noi = neg(ray.o * inv);
            fma_ok = __builtin_fmaxf(__builtin_fabsf(noi.x), __builtin_fmaxf(__builtin_fabsf(noi.y), __builtin_fabsf(noi.z))) <= PT_FAST_BOX_LIMIT;
#endif
            regular = regular_ray(inv, ray.o, a, sc.bounded != 0);
            if (EXT && N1OFF == PT_OBJ_N1_OFFSET) obj_cache_fill(sc, obj_at, ray.o, ray.d, xf_lds);
            h.t = T_MAX; h.prim = -1;
            code = (new_path && regular) ? primary_code : sc.root_code;
        }
        new_path = false; new_walk = false;
        PT_TIME(5);

        // walking waves issue ahead of shading ones (short dependent steps first: -1.5 %)
        __builtin_amdgcn_s_setprio(1);
        // every ray of the wave REGULAR (pt_device.hpp)?  Re-evaluated once per SHADE step; a lane
        // whose irregular walk is over keeps the wave on the general slab test until its next walk
        const bool all_regular = __ballot(!regular) == 0ull;
#ifdef PT_FAST_MODE
        const bool all_fma_ok = __ballot(!fma_ok) == 0ull;
#endif
        bool done = false;
        for (;;) {
            const int cS = lanes((uint32_t)code < (uint32_t)n_nodes), cL = lanes(code < 0);
            const int cZ = lanes(code == n_nodes);                               // walk ended: to be shaded
            const int vS = cS * (EXT ? PT_W_SPLIT_EXT : PT_W_SPLIT), vL = cL * (EXT ? PT_W_LEAF_EXT : PT_W_LEAF), vZ = cZ * (EXT ? PT_W_SHADE_EXT : PT_W_SHADE);
            PT_TIME(4);
            if (vS >= vL && vS >= vZ) {
                if (vS == 0) { done = true; break; }                             // no lane has work left
                // SPLIT run: the lanes at a Split stay in the loop (exec shrinks as they leave the
                // state); the run ends for all of them once fewer than `keep` are left
                const int keep = (cS * (EXT ? PT_SPLIT_KEEP_EXT : PT_SPLIT_KEEP) + 7) >> 3;
                if ((uint32_t)code < (uint32_t)n_nodes) {
                    if (all_regular) {
#if defined(PT_FAST_MODE) && PT_FAST_BOX
                        if (all_fma_ok) code = split_run_regular_fast<ALL_IN_LDS, N1OFF>(code, inv, noi, h.t, sc, n_nodes, keep);
                        else code = split_run_regular<ALL_IN_LDS, N1OFF>(code, ray.o, inv, h.t, sc, n_nodes, keep);
#elif !defined(PT_STATS)
                        code = split_run_regular<ALL_IN_LDS, N1OFF>(code, ray.o, inv, h.t, sc, n_nodes, keep);
#else   // the same run in C++, with its step counters
                        for (;;) {
                            PT_STAT(0, lanes(true));
                            float4 n0, n1;
                            fetch(code, n0, n1, true);
                            code = split_step<true>(n0, n1, ray.o, inv, h.t);
                            const bool at = (uint32_t)code < (uint32_t)n_nodes;
                            const bool few = lanes(at) < keep;
                            if ((int)!at | (int)few) break;
                        }
#endif
                    } else {
                        for (;;) {
                            PT_STAT(0, lanes(true));
                            float4 n0, n1;
                            fetch(code, n0, n1, true);
                            code = split_step<false>(n0, n1, ray.o, inv, h.t);
                            const bool at = (uint32_t)code < (uint32_t)n_nodes;
                            const bool few = lanes(at) < keep;
                            if ((int)!at | (int)few) break;
                        }
                    }
                }
                PT_TIME(0);
            } else if (vL >= vZ) {
                // LEAF run: one Sphere::trace per iteration for the lanes standing on a Leaf
                // (extension scenes: small trees, dear rect tests -- a run goes on while half of its starters remain: -3 %)
                const int keepL = (cL * (EXT ? PT_LEAF_KEEP_EXT : PT_LEAF_KEEP) + 7) >> 3;
                if (code < 0) {
                    for (;;) {
                        PT_STAT(1, lanes(true));
                        float4 n0, n1;
                        fetch(code, n0, n1);
                        if (EXT && (__float_as_int(n1.y) & 1)) {
                            if (N1OFF == PT_OBJ_N1_OFFSET) rect_step_cached(n0, n1, obj_at, h);
                            else rect_step(sc, n0, n1, ray.o, ray.d, h, xf_lds);
                        }
                        else leaf_step(n0, n1, ray.o, ray.d, a, ra, PT_FAST_DIV && all_regular, h);
                        code = __float_as_int(n1.w);
                        const bool at = code < 0;
                        const bool few = lanes(at) < keepL;
                        if ((int)!at | (int)few) break;
                    }
                }
                PT_TIME(1);
            } else break;
        }
        if (done) break;

        __builtin_amdgcn_s_setprio(0);
        // SHADE: the lanes whose walk has ended.  A hit scatters (one iteration of
        // `for _ in 0..MAX_TRACES`, raytracer.rs:52-61) and usually starts the next walk; a miss,
        // an absorbed path or the bounce cap ends the path in the same step.
        {
            const bool walked = code == n_nodes;
            bool ended = walked && h.prim < 0, to_sky = true;
            PT_STAT(2, lanes(walked && h.prim >= 0));
            if (walked && h.prim >= 0) {
                f3 am; Ray nr;
                bool scattered;
                if (EXT) {
                    f3 emitted; bool light;
                    scattered = scatter_ext(sc, ray, h, rng, am, nr, emitted, light, xf_lds);
                    if (light) rad = rad + emitted * att;                        // Material::emitted, before scatter's verdict
                } else {
                    scattered = scatter(sc, ray, h, rng, am, nr);
                }
                if (scattered) {
                    att = att * am;                                              // raytracer.rs:57
                    ray = nr;
                    bounce++;
                    bool killed = false;
                    if (EXT && rp.rr_start >= 0 && bounce >= rp.rr_start) {      // Russian roulette on the largest component
                        const float q = __builtin_fmaxf(att.x, __builtin_fmaxf(att.y, att.z));
                        if (q < 1.0f) {
                            const float r = rng_f32(rng);
                            if (!(r < q)) killed = true;
                            else att = att / q;
                        }
                    }
                    if (killed) { ended = true; to_sky = false; }
                    else if (bounce >= (EXT ? rp.max_bounces : 5)) ended = true; // MAX_TRACES, raytracer.rs:48
                    else new_walk = true;
                } else {
                    ended = true; to_sky = false;                                // raytracer.rs:60: return 0
                }
            }
            PT_TIME(2);
            PT_STAT(3, lanes(ended));
            if (ended) {
                f3 color = EXT ? rad : fill(0.0f);
                if (to_sky) {
                    if (!EXT) color = sky(ray, att);                             // raytracer.rs:67-70
                    else if (rp.background_kind == 0) color = rad + sky(ray, att);
                    else color = rad + mk(rp.bg_r, rp.bg_g, rp.bg_b) * att;
                }
                sum = sum + color;                                               // raytracer.rs:88
                s++;
                sxy++;
                if ((sxy & 0xffffu) == (uint32_t)S) sxy += 0x10000u - (uint32_t)S;
                new_path = true;
            }
            PT_TIME(3);
        }
    }

#ifdef PT_STATS
    if (lane == 0 && rp.stats)
        for (int k = 0; k < 4; k++) { atomicAdd(&rp.stats[2 * k], (unsigned long long)st_n[k]); atomicAdd(&rp.stats[2 * k + 1], (unsigned long long)st_l[k]); }
    if (lane == 0 && rp.stats)
        for (int k = 0; k < 6; k++) atomicAdd(&rp.stats[8 + k], st_c[k]);
#endif
    // What the epilogue needs of the item's first lines -- the pixel's address, the lane's slot and chunk -- is formed
    // again from values the loop keeps anyway (behind an empty asm, so that the compiler cannot reuse the first
    // computation): held across the loop they were seven spilled registers per lane, written at the start of every
    // item and read back at its end, 1.2 GB of private-segment traffic per headline frame.
    uint32_t pixel_e = pixel;
    int lane_e = (int)(threadIdx.x & 63u);
    asm volatile("" : "+v"(pixel_e), "+v"(lane_e));
    const bool valid_e = pixel_e < (uint32_t)(cam.width * cam.height) && px < cam.width;
    if (L == 1) {
        if (valid_e) { radiance[3 * pixel_e] = sum.x; radiance[3 * pixel_e + 1] = sum.y; radiance[3 * pixel_e + 2] = sum.z; }
    } else {
        // lanes_per_pixel > 1: the pixel's chunk sums, added in chunk order by its chunk-0 lane
        const int slot_e = lane_e % P, chunk_e = lane_e / P;
        f3 total = fill(0.0f);
        if (resume && chunk_e == 0 && valid_e)
            total = mk(radiance[3 * pixel_e], radiance[3 * pixel_e + 1], radiance[3 * pixel_e + 2]);
        for (int k = 0; k < L; k++) {
            const int src = k * P + slot_e;
            total = total + mk(__shfl(sum.x, src), __shfl(sum.y, src), __shfl(sum.z, src));
        }
        if (chunk_e == 0 && valid_e) {
            radiance[3 * pixel_e] = total.x; radiance[3 * pixel_e + 1] = total.y; radiance[3 * pixel_e + 2] = total.z;
        }
    }
    if (sched.n_pieces > 1) {
        // publish this piece's sums for the wave that draws the tile's next piece
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&sched.done[tile], piece + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// EXT: the instantiation for scenes uploaded with pt_upload_scene_ext (rects, instances, lights, integrator
// options).  The reference's scenes always run EXT = false, whose code is unchanged.
// Stages the group's LDS copy of the nodes (and transforms); returns the LDS address of the transform table.
template <bool EXT, int N1OFF>
__device__ __forceinline__ bool stage_group(const Scene &sc, const SchedParams &sched, uint32_t &xf_lds, int &ticket_slot)
{
    // dynamic LDS: n0[0 .. lds_nodes), n1[0 .. lds_nodes) from byte N1OFF, then the group's ticket.  There is no
    // static __shared__ in this kernel, so the array starts at LDS address 0 and 16 * slot IS the LDS address of
    // a node's n0 entry: no base add per node visit (checked here: error flag 2).
    extern __shared__ float4 lds_n0[];
    if ((uint32_t)(uintptr_t)lds_n0 != 0u) { if (threadIdx.x == 0) sched.ctrl[1] = 2; return false; }
    const int lds_nodes = sc.lds_nodes;
    if (lds_nodes * 16 > N1OFF) { if (threadIdx.x == 0) sched.ctrl[1] = 2; return false; }   // launch_path never asks for this
    ticket_slot = N1OFF / 16 + lds_nodes;
    for (int i = threadIdx.x; i < lds_nodes; i += blockDim.x) {
        lds_n0[i] = sc.n0[i];
        lds_n0[N1OFF / 16 + i] = sc.n1[i];
    }
    // extension scenes: the instance transforms (two float4 each) behind the ticket word
    const int xf_slot = ticket_slot + 1;
    xf_lds = EXT && sc.lds_xforms > 0 ? (uint32_t)xf_slot * 16u : XF_GLOBAL;
    if (EXT)
        for (int i = threadIdx.x; i < 2 * sc.lds_xforms; i += blockDim.x) lds_n0[xf_slot + i] = sc.xforms[i];
    return true;
}
// bytes of dynamic LDS stage_group fills
static size_t group_lds_bytes(const Scene &sc, int n1off)
{
    return (size_t)n1off + ((size_t)sc.lds_nodes + 1) * sizeof(float4) + (sc.ext ? 2 * (size_t)sc.lds_xforms * sizeof(float4) : 0);
}

// The launch's work-distribution words clean up after themselves: a launch needs the ticket counter and the tiles'
// `done` words at zero, and instead of a memset in front of every launch (one more operation on the stream -- and one
// more node for a HIP graph to order) the LAST workgroup to finish puts them back.  Every workgroup counts itself
// out in ctrl[2] once all its waves are through; whoever brings the count to gridDim.x knows that no wave of the
// launch will read a word again.  The error word ctrl[1] is left alone (the host reads and clears it).  The words
// start at zero: the library's static array is zero-initialised, a context's own scratch is cleared when allocated.
__device__ __forceinline__ void retire_group(const SchedParams &sched)
{
    // (no static __shared__ in these kernels -- the staged nodes must start at LDS address 0 -- so the flag
    // takes the first word of the staged copy, which nobody reads after the barrier)
    extern __shared__ float4 lds_n0[];
    int &last = *(int *)&lds_n0[0];
    __syncthreads();                                                             // all waves of the group are through their items
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(&sched.ctrl[2], 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (sched.n_pieces > 1)
        for (int i = threadIdx.x; i < sched.n_tiles; i += blockDim.x) sched.done[i] = 0;
    if (threadIdx.x == 0) { sched.ctrl[0] = 0; sched.ctrl[2] = 0; }
}

// The launch without the item loop: one workgroup of 4 waves per 4 items.  It serves the scenes of a handful of
// primitives only (launch_path).
template <bool EXT = false>
__global__ __launch_bounds__(256, PT_MIN_WAVES) void path_kernel(Scene sc, Camera cam, RenderParams rp, SchedParams sched, float *radiance)
{
    // One ticket per workgroup, drawn while the nodes are staged: the group's waves take the
    // (piece, tile) items 4*ticket .. 4*ticket+3.  Tickets follow the order in which groups
    // START, so the item a wave may have to wait for (same tile, previous piece: a lower
    // item) belongs to a group that is already running or done -- whatever order the
    // hardware dispatches blockIdx in.
    extern __shared__ float4 lds_n0[];
    uint32_t xf_lds; int ticket_slot;
    if (!stage_group<EXT, PT_SMALL_N1_OFFSET>(sc, sched, xf_lds, ticket_slot)) return;
    int &ticket = *(int *)&lds_n0[ticket_slot];
    if (threadIdx.x == 0) ticket = atomicAdd(&sched.ctrl[0], 1);
    __syncthreads();
    const int n_items = sched.n_tiles * sched.n_pieces;
    const int item = __builtin_amdgcn_readfirstlane(ticket) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (item < n_items) render_item<true, EXT, PT_SMALL_N1_OFFSET>(sc, cam, rp, sched, radiance, item, xf_lds);
    retire_group(sched);
}

// The persistent launch: as many workgroups as the chip holds, each wave drawing items until none are left.  The
// item's code is a real function call, so that the loop around it takes none of its registers (inlined, the item
// loop cost 6 % in spilled scalar registers: profiles/r01).
struct PathArgs { Scene sc; Camera cam; RenderParams rp; SchedParams sched; float *radiance; };
// The arguments travel as the address of the kernel's own argument block (constant address space, so the
// function reads them with scalar loads): __builtin_amdgcn_kernarg_segment_ptr() is NULL inside a callee -- the
// kernel has to hand the pointer over.  Function arguments arrive in vector registers: made uniform again first.
typedef __attribute__((address_space(4))) const PathArgs *PathArgsPtr;
// not_tail_called: the function uses every vector register the launch bounds allow, sixteen of them callee-saved by
// the ABI (v40-47, v56-63) -- saved to the wave's private segment on entry and restored on exit, 9 KB per call that
// L2 has evicted by the time the item is done: 4.8 GB of HBM traffic per headline frame for values nobody holds (the
// loop around the call keeps three).  LLVM drops a callee's saves when it can see every caller (internal linkage,
// no recursion, address never taken: this function) -- unless a call site carries the `tail` marker, which the
// optimiser puts on any call that cannot reach the caller's stack, this one included.  The attribute keeps the marker
// off; the caller then saves what IT holds across the call (two registers).
template <bool ALL_IN_LDS, bool EXT, int N1OFF>
__device__ __attribute__((noinline, not_tail_called)) void render_item_call(const uint32_t args_lo, const uint32_t args_hi, const int item_v, const uint32_t xf_lds_v)
{
#if __HIP_DEVICE_COMPILE__   // the host pass has no address spaces to copy from
    const uint64_t addr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)args_hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)args_lo);
    const PathArgsPtr a = (PathArgsPtr)addr;
    const int item = __builtin_amdgcn_readfirstlane(item_v);
    const uint32_t xf_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)xf_lds_v);
    const Scene sc = a->sc;
    const Camera cam = a->cam;
    const RenderParams rp = a->rp;
    const SchedParams sched = a->sched;
    render_item<ALL_IN_LDS, EXT, N1OFF>(sc, cam, rp, sched, a->radiance, item, xf_lds);
#else
    (void)args_lo; (void)args_hi; (void)item_v; (void)xf_lds_v;
#endif
}
template <bool ALL_IN_LDS, bool EXT = false, int N1OFF = PT_LDS_N1_OFFSET>
__global__ __launch_bounds__(64 * WAVES_PER_GROUP, PT_MIN_WAVES) void path_kernel_persistent(PathArgs a)
{
    uint32_t xf_lds; int ticket_slot;
    // The launch's own clock, for the bench record: workgroup 0 lives as long as the launch does, and its first wave
    // stamps the shader-cycle counter and the constant 100 MHz counter ONCE at its start and once at its end (two stamps
    // per launch; no stamp executes inside an item).  The four words they go to are read by the host only.
    const bool stamper = blockIdx.x == 0 && threadIdx.x < 64;
    unsigned long long clk0 = 0, ref0 = 0;
    if (stamper) { clk0 = __builtin_amdgcn_s_memtime(); ref0 = __builtin_amdgcn_s_memrealtime(); }
    if (!stage_group<EXT, N1OFF>(a.sc, a.sched, xf_lds, ticket_slot)) return;
    __syncthreads();
    const int n_items = a.sched.n_tiles * a.sched.n_pieces;
    const uint64_t args = (uint64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();   // PathArgs is the kernel's only argument: offset 0
#ifdef PT_PHASES
    unsigned long long t_draw = 0, t_item = 0;
#endif
    for (;;) {
        int item = 0;
#ifdef PT_PHASES
        const unsigned long long t0 = clock64();
#endif
        if ((threadIdx.x & 63) == 0) item = atomicAdd(&a.sched.ctrl[0], 1);
        item = __builtin_amdgcn_readfirstlane(item);
#ifdef PT_PHASES
        const unsigned long long t1 = clock64();
        t_draw += t1 - t0;
#endif
        if (item >= n_items) break;
        render_item_call<ALL_IN_LDS, EXT, N1OFF>((uint32_t)args, (uint32_t)(args >> 32), item, xf_lds);
#ifdef PT_PHASES
        t_item += clock64() - t1;
#endif
    }
    if (stamper && threadIdx.x == 0) {
        const unsigned long long clk = __builtin_amdgcn_s_memtime() - clk0, ref = __builtin_amdgcn_s_memrealtime() - ref0;
        a.sched.ctrl[4] = (int32_t)(uint32_t)clk; a.sched.ctrl[5] = (int32_t)(uint32_t)(clk >> 32);
        a.sched.ctrl[6] = (int32_t)(uint32_t)ref; a.sched.ctrl[7] = (int32_t)(uint32_t)(ref >> 32);
    }
    retire_group(a.sched);
#ifdef PT_PHASES
    if ((threadIdx.x & 63) == 0) {
        atomicAdd((unsigned long long *)&a.sched.ctrl[16], t_draw);
        atomicAdd((unsigned long long *)&a.sched.ctrl[18], t_item);
        atomicAdd((unsigned long long *)&a.sched.ctrl[20], 1ull);
    }
#endif
}

#ifndef PT_FAST_MODE   // everything from here to the launchers is built once, in the EXACT translation unit
// K1: primitive id (and t) hit by the primary ray of one sample index.
template <bool EXT>
__global__ __launch_bounds__(64) void primary_hit_kernel(Scene sc, Camera cam, int S, int sample, int32_t *__restrict__ hit_id,
                                                         float *__restrict__ hit_t)
{
    const int lane = threadIdx.x;
    const int tiles_x = (cam.width + 7) / 8;
    const int px = (blockIdx.x % tiles_x) * 8 + (lane & 7);
    const int py = (blockIdx.x / tiles_x) * 8 + (lane >> 3);
    if (px >= cam.width || py >= cam.height) return;
    const Ray ray = sample_ray(cam, S, px, py, sample);
    const Hit h = scene_trace<EXT>(sc, ray);
    const int pixel = py * cam.width + px;
    if (hit_id) hit_id[pixel] = h.prim >= 0 ? sc.prim_ids[h.prim] : -1;   // device index (rank) -> the caller's id
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

// The primary-ray candidate lists (Scene::cand): one lane per 8x8 pixel block of the frame.  No counterpart in the
// reference, whose every ray walks the tree from the root (scene/mod.rs:24-34); what a list holds is the set of
// Leaves that walk can end on for the camera rays of the block (raytracer.rs:77-85 + camera.rs:37-44).
//
// The block's camera rays leave the eye inside the cone over a parallelogram of directions -- dir(u, v) is
// affine in (u, v), u in [x0, x0 + 8] / W, v in 1 - [y0, y0 + 8] / H (a pixel's sample offsets run over [0, 1]).
// A ray inside the cone can only meet a sphere whose centre lies within its radius of the cone, hence within
// its radius of the inner half-space of each of the cone's four side planes.  A sphere is LEFT OUT only on proof:
// its centre is farther than radius + margin outside one of the planes, evaluated in fp64, where the margin
// covers, a hundred times over, everything fp32 does to the kernel's own rays and tests:
//   * the direction the kernel computes differs from the affine one by at most ~1e-6 (|llc| + |horizontal| +
//     |vertical| + |origin|) in each component (four roundings of terms of that size, and u, v carry 2^-24 relative
//     errors of their own): an angle of at most that over the distance to the image plane -- `theta`, taken 10 x larger;
//   * Sphere::trace's discriminant b^2 - a c carries an absolute error of a few 2^-24 |oc|^2 |d|^2, so it can come out
//     positive for a ray that passes the centre at up to sqrt(r^2 + 5e-7 D^2) <= r + 2.5e-7 D^2 / r: taken as
//     r (1 + 1e-4) + 1e-5 D^2 / r (a vanishing radius keeps its sphere in every list).
// Every comparison is written so that a NaN keeps the sphere, and a camera whose cone cannot be formed (degenerate
// or non-finite fields) marks the block "walk the tree".  The lists come out in rank order.
__global__ __launch_bounds__(64) void primary_list_kernel(Scene sc, Camera cam, const int32_t *__restrict__ leaf_codes, int n_leaves,
                                                          int blocks_x, int blocks_y, int32_t *__restrict__ lists)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= blocks_x * blocks_y) return;
    int32_t *out = lists + (size_t)t * PT_PRIMARY_K;
    const int bx = t % blocks_x, by = t / blocks_x;
    struct d3 { double x, y, z; };
    auto D3 = [](f3 a) { d3 r = {(double)a.x, (double)a.y, (double)a.z}; return r; };
    auto dotd = [](d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
    auto crossd = [](d3 a, d3 b) { d3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; return r; };
    auto norm1 = [](d3 a) { return __builtin_fabs(a.x) + __builtin_fabs(a.y) + __builtin_fabs(a.z); };
    const d3 llc = D3(cam.llc), o = D3(cam.origin), H = D3(cam.horizontal), V = D3(cam.vertical);
    const double u0 = (double)(bx * 8) / (double)cam.width, u1 = (double)(bx * 8 + 8) / (double)cam.width;
    const double v1 = 1.0 - (double)(by * 8) / (double)cam.height, v0 = 1.0 - (double)(by * 8 + 8) / (double)cam.height;
    auto corner = [&](double u, double v) { d3 r = {llc.x + H.x * u + V.x * v - o.x, llc.y + H.y * u + V.y * v - o.y, llc.z + H.z * u + V.z * v - o.z}; return r; };
    const d3 c[4] = {corner(u0, v0), corner(u1, v0), corner(u1, v1), corner(u0, v1)};
    const d3 centre = {c[0].x + c[1].x + c[2].x + c[3].x, c[0].y + c[1].y + c[2].y + c[3].y, c[0].z + c[1].z + c[2].z + c[3].z};
    // the image plane's distance from the eye bounds |dir| from below over the whole block
    d3 ip = crossd(H, V);
    const double ipl = __builtin_sqrt(dotd(ip, ip));
    const double focal = __builtin_fabs(dotd(ip, c[0])) / ipl;
    const double theta = 1e-5 * (norm1(llc) + norm1(H) + norm1(V) + norm1(o)) / focal;
    bool cone_ok = theta < 1e-2 && ipl > 0.0;                                   // false for NaN as well
    d3 n[4];
    for (int k = 0; k < 4; k++) {
        d3 m = crossd(c[k], c[(k + 1) & 3]);
        const double len = __builtin_sqrt(dotd(m, m));
        double sgn = dotd(m, centre);
        cone_ok = cone_ok && len > 0.0 && __builtin_isfinite(len) && (sgn > 0.0 || sgn < 0.0);
        sgn = sgn < 0.0 ? -1.0 / len : 1.0 / len;                               // inward unit normal
        n[k].x = m.x * sgn; n[k].y = m.y * sgn; n[k].z = m.z * sgn;
    }
    if (!cone_ok) { out[0] = 1; return; }
    int count = 0;
    for (int i = 0; i < n_leaves; i++) {
        const int32_t code = leaf_codes[i];
        const float4 s = sc.n0[code_index(code)];                               // (centre, radius^2)
        const d3 oc = {(double)s.x - o.x, (double)s.y - o.y, (double)s.z - o.z};
        const double r = __builtin_sqrt((double)s.w), D2 = dotd(oc, oc);
        const double reach = r * (1.0 + 1e-4) + 1e-5 * D2 / r + theta * __builtin_sqrt(D2);
        bool out_of_cone = false;
        for (int k = 0; k < 4; k++) out_of_cone = out_of_cone || dotd(n[k], oc) < -reach;   // NaN: not proven outside
        if (out_of_cone) continue;
        if (count == PT_PRIMARY_K) { out[0] = 1; return; }
        out[count++] = code;
    }
    for (int k = count; k < PT_PRIMARY_K; k++) out[k] = 0;                      // the path kernel reads all PT_PRIMARY_K entries
}

// pt_selftest_unary: one fp32 bit pattern per lane
template <int FN>
__global__ __launch_bounds__(256) void unary_selftest_kernel(unsigned long long *count, uint32_t base)
{
    const uint32_t bits = base + blockIdx.x * 256u + threadIdx.x;
    const float x = __uint_as_float(bits);
    float ref, got;
    if (FN == 0) { ref = 1.0f / x; got = p_rcp(x); }
    else if (FN == 1) { ref = __builtin_sqrtf(x); got = p_sqrt(x); }
    else { ref = 1.0f / __builtin_sqrtf(x); got = rsq(x); }
    const bool same = (ref != ref && got != got) || __float_as_uint(ref) == __float_as_uint(got);
    const unsigned long long bad = __ballot(!same);
    if (bad != 0ull && (threadIdx.x & 63) == 0) atomicAdd(count, (unsigned long long)__popcll(bad));
}

// pt_selftest_grid: lane = (pixel coordinate p, grid coordinate g); axis 0 = columns (u), 1 = rows (v)
__global__ __launch_bounds__(256) void grid_selftest_kernel(unsigned long long *count, int extent, int S, float inv_s1,
                                                            float inv_extent, int axis)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    bool bad = false;
    if (i < (long long)extent * S) {
        const int p = (int)(i / S), g = (int)(i % S);
        const float s1 = (float)(S - 1);
        const float ref_f = (float)p + ((float)g / s1), got_f = (float)p + div_by((float)g, s1, inv_s1);
        float ref = ref_f / (float)extent, got = div_by(got_f, (float)extent, inv_extent);
        if (axis == 1) { ref = 1.0f - ref; got = 1.0f - got; }
        bad = __float_as_uint(ref_f) != __float_as_uint(got_f) || __float_as_uint(ref) != __float_as_uint(got);
    }
    const unsigned long long m = __ballot(bad);
    if (m != 0ull && (threadIdx.x & 63) == 0) atomicAdd(count, (unsigned long long)__popcll(m));
}

#endif   // !PT_FAST_MODE
}  // namespace

#ifndef PT_FAST_MODE
// pt_selftest_division: 64 operand pairs (x, a) per lane for Sphere::trace's division x / a through
// ra = RN(1/a) (pt_device.hpp div_by, regular_ray): a over the whole REGULAR range [2^-32, 2^32]
// with mantissas biased to the ends of the binade; x random (mode 0) or built so that x / a lies
// within two ulps of a rounding midpoint (mode 1), the case a division algorithm gets wrong first;
// mode 2 pins a's mantissa to all ones (dot(d, d) of a normalised direction is 1 - 2^-24 more often than not).
__device__ __forceinline__ uint64_t selftest_mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void division_selftest_kernel(unsigned long long *count, uint64_t first, int mode)
{
    const uint64_t id = first + (uint64_t)blockIdx.x * 256u + threadIdx.x;
    unsigned bad = 0;
    for (int k = 0; k < 64; k++) {
        const uint64_t h = selftest_mix(id * 64 + (uint64_t)k);
        uint32_t am = (uint32_t)h & 0x7fffffu;
        if ((h >> 60) == 0) am = 0x7fffffu - (am & 0xffu);
        if ((h >> 60) == 1) am = am & 0xffu;
        if (mode == 2) am = 0x7fffffu;                                           // a = (2 - 2^-23) 2^e: dot(d, d) of a normalised d
        int ae = (int)((h >> 23) % 65) - 32;
        if (ae == 32) am = 0;                                                    // the range ends AT 2^32
        const float a = __uint_as_float(((uint32_t)(ae + 127) << 23) | am);
        const uint64_t g = selftest_mix(h);
        float x;
        if (mode == 0 || (mode == 2 && (g >> 61) == 0)) {
            const uint32_t xm = (uint32_t)g & 0x7fffffu;
            const int xe = (int)((g >> 23) % 107) - 53;                          // |x| < 2^53 (regular_ray)
            x = __uint_as_float(((uint32_t)(g >> 63) << 31) | ((uint32_t)(xe + 127) << 23) | xm);
        } else {
            const uint32_t qm = (uint32_t)g & 0x7fffffu;
            const int qe = (int)((g >> 23) % 40) - 9;                            // quotient in [2^-9, 2^30]: around and above T_MIN
            const double q = (double)__uint_as_float(((uint32_t)(qe + 127) << 23) | qm);
            const double mid = q + (double)__uint_as_float((uint32_t)(qe + 127 - 24) << 23);   // half an ulp above q
            const float x0 = (float)((double)a * mid);
            x = __uint_as_float(__float_as_uint(x0) + (uint32_t)((int)((g >> 40) % 5) - 2));
            if ((g >> 62) & 1) x = -x;
        }
        const float ref = x / a, got = div_by(x, a, rcp_core(a));
        if (__float_as_uint(ref) != __float_as_uint(got) && !(ref != ref && got != got)) bad++;
    }
    if (bad) atomicAdd(count, (unsigned long long)bad);
}


// pt_clock_probe: what the shader clock is RIGHT NOW.  A frame's time is cycles / clock, and the clock is the
// chip's own choice under load (power, temperature, the box): a bench record that carries it tells a slower box
// from a slower kernel.  Every SIMD gets 8 waves (the path kernel's occupancy) of a dependent chain of un-fused
// fp32 multiplies for about two milliseconds; wave 0 of each group stamps s_memtime (shader cycles) and
// s_memrealtime (a constant 100 MHz) around it, into a buffer of its own.  Not the path kernel: no stamp executes
// in that one.
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long *stamps, float *sink, int iters)
{
    float a0 = 1.0f + (float)threadIdx.x * 1e-6f, a1 = a0 + 1e-3f, a2 = a0 + 2e-3f, a3 = a0 + 3e-3f;
    const float m = 1.0000001f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++)
            asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    if (a0 + a1 + a2 + a3 == 12345.678f) sink[0] = a0;   // keeps the chain alive; never true in practice, and no output is computed from a stamp
}

#endif   // !PT_FAST_MODE

// Scratch of pt_render_device_static (include/pt_amd.h): the work-distribution words of one
// path-kernel launch (SchedParams: 64 control words + one `done` word per 8x8 tile), for callers
// that cannot allocate.  Each GPU the library runs on holds its own copy of this array until the
// process ends; host code never uses the array's name as an address -- ptk::static_sched resolves the
// device address once per device and the launches take that pointer like any other buffer.
#ifndef PT_FAST_MODE
constexpr int kStaticSchedTiles = 1 << 18;                  // 8x8 tiles: frames up to 16.7 M pixels (a 4K frame has 129,600)
constexpr int kStaticSchedWords = 64 + kStaticSchedTiles;   // 1 MiB + 256 B
static __device__ __align__(256) int32_t scratch[kStaticSchedWords];
#endif   // !PT_FAST_MODE

namespace PTK {

#ifndef PT_FAST_MODE

// Device address and size of `scratch` on `device`, which must be the CURRENT device (pt_create
// calls this right after hipSetDevice).  Resolved once per device and kept: every GPU holds its own
// copy of the array at its own address.  Next to the address the table keeps the array's IN-FLIGHT
// record: all contexts of a GPU share the array, so "one launch at a time" is a property of the
// array, not of a context (static_claim / static_mark below).
namespace {
constexpr int kMaxDevices = 64;
struct StaticSlot {
    int32_t *ptr = nullptr;
    hipEvent_t done = nullptr;     // recorded behind the most recent launch that used the array
    hipStream_t stream = nullptr;  // the stream that launch went to
    bool used = false;
};
std::mutex g_static_lock;
StaticSlot g_static[kMaxDevices];
}  // namespace

hipError_t static_sched(int device, int32_t **out, size_t *bytes)
{
    *bytes = sizeof(int32_t) * (size_t)kStaticSchedWords;
    std::lock_guard<std::mutex> guard(g_static_lock);
    if (device >= 0 && device < kMaxDevices && g_static[device].ptr) { *out = g_static[device].ptr; return hipSuccess; }
    void *p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (device >= 0 && device < kMaxDevices) g_static[device].ptr = (int32_t *)p;
    *out = (int32_t *)p;
    return hipSuccess;
}

// May a launch on `stream` use the device's scratch array now?  *busy = the array's previous launch went to
// ANOTHER stream and has not completed (launches on one stream are ordered; a launch on a second stream would
// clear the tickets under the running kernel).  `device` must be current.  Allocates nothing after the first
// call on a device (which creates the event).
hipError_t static_claim(int device, hipStream_t stream, bool *busy)
{
    *busy = false;
    if (device < 0 || device >= kMaxDevices) return hipSuccess;   // beyond the table: unguarded, as documented for such devices
    std::lock_guard<std::mutex> guard(g_static_lock);
    StaticSlot &s = g_static[device];
    if (!s.done) {
        const hipError_t e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
        if (e != hipSuccess) { s.done = nullptr; return e; }
    }
    if (s.used && s.stream != stream && hipEventQuery(s.done) == hipErrorNotReady) *busy = true;
    return hipSuccess;
}
// records the launch just enqueued on `stream` as the array's most recent one
hipError_t static_mark(int device, hipStream_t stream)
{
    if (device < 0 || device >= kMaxDevices) return hipSuccess;
    std::lock_guard<std::mutex> guard(g_static_lock);
    StaticSlot &s = g_static[device];
    if (!s.done) return hipSuccess;
    const hipError_t e = hipEventRecord(s.done, stream);
    if (e != hipSuccess) return e;
    s.stream = stream;
    s.used = true;
    return hipSuccess;
}

hipError_t launch_primary_lists(const Scene &sc, const Camera &cam, const int32_t *d_leaf_codes, int n_leaves, int32_t *d_lists,
                                hipStream_t stream)
{
    const int bx = (cam.width + 7) / 8, by = (cam.height + 7) / 8;
    hipLaunchKernelGGL(primary_list_kernel, dim3((unsigned)((bx * by + 63) / 64)), dim3(64), 0, stream, sc, cam, d_leaf_codes, n_leaves, bx,
                       by, d_lists);
    return hipGetLastError();
}

hipError_t launch_clock_probe(unsigned long long *d_stamps, float *d_sink, int groups, int iters, hipStream_t stream)
{
    hipLaunchKernelGGL(clock_probe_kernel, dim3((unsigned)groups), dim3(256), 0, stream, d_stamps, d_sink, iters);
    return hipGetLastError();
}

hipError_t launch_division_selftest(int log2_pairs, unsigned long long *d_count, hipStream_t stream)
{
    // 2^log2_pairs operand pairs in modes 0 (random x) and 1 (x / a next to a rounding midpoint), half each,
    // and half as many again in mode 2: a's mantissa pinned to all ones (1/8 random x, 7/8 next to midpoints); 64 per lane
    const uint64_t lanes = (1ull << log2_pairs) / 64 / 2;
    for (int mode = 0; mode < 3; mode++)
        for (uint64_t first = 0; first < lanes; first += (1ull << 28)) {
            const uint64_t n = lanes - first < (1ull << 28) ? lanes - first : (1ull << 28);
            hipLaunchKernelGGL(division_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_count,
                               first + ((uint64_t)mode << 60), mode);
        }
    return hipGetLastError();
}

hipError_t launch_grid_selftest(int width, int height, int S, float inv_s1, float inv_w, float inv_h, unsigned long long *d_count,
                                hipStream_t stream)
{
    const long long nu = (long long)width * S, nv = (long long)height * S;
    hipLaunchKernelGGL(grid_selftest_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, stream, d_count, width, S, inv_s1, inv_w, 0);
    hipLaunchKernelGGL(grid_selftest_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, stream, d_count, height, S, inv_s1, inv_h, 1);
    return hipGetLastError();
}

hipError_t launch_unary_selftest(int fn, unsigned long long *d_count, hipStream_t stream)
{
    for (uint64_t base = 0; base < (1ull << 32); base += (1ull << 28)) {
        const dim3 grid((1u << 28) / 256u), block(256);
        if (fn == 0) hipLaunchKernelGGL(unary_selftest_kernel<0>, grid, block, 0, stream, d_count, (uint32_t)base);
        else if (fn == 1) hipLaunchKernelGGL(unary_selftest_kernel<1>, grid, block, 0, stream, d_count, (uint32_t)base);
        else hipLaunchKernelGGL(unary_selftest_kernel<2>, grid, block, 0, stream, d_count, (uint32_t)base);
    }
    return hipGetLastError();
}

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

// control words + one `done` word per 8x8 tile (the only tiling that is split into pieces)
bool launch_is_persistent(const Scene &sc) { return sc.ext || sc.n_nodes >= PT_SMALL_TREE; }
size_t sched_bytes(const Camera &cam) { return (64 + (size_t)tiles_for(cam, 1)) * sizeof(int32_t); }
#else
using ptk::launch_is_persistent;
using ptk::sched_bytes;
using ptk::tiles_for;
#endif   // !PT_FAST_MODE

hipError_t launch_path(const Scene &sc, const Camera &cam, const RenderParams &rp, int32_t *d_sched, float *d_radiance,
                       hipStream_t stream)
{
    SchedParams sp;
    sp.n_tiles = tiles_for(cam, rp.lanes);
    sp.ctrl = d_sched;
    sp.done = d_sched + 64;
    // enough items that the last ones are a small fraction of a wave slot's share of the frame
    // (256 CUs x 32 wave slots x 64 items); whole-pixel lanes only, at least PT_PIECE_MIN samples per piece
    const int count = rp.s1 - rp.s0;
    int pieces = 1;
    if (rp.lanes == 1 && count > PT_PIECE_MIN) {
        const long long want = 256ll * 32 * PT_ITEMS_PER_SLOT;
        pieces = (int)((want + sp.n_tiles - 1) / sp.n_tiles);
        if (pieces > count / PT_PIECE_MIN) pieces = count / PT_PIECE_MIN;
        if (pieces < 1) pieces = 1;
    }
    sp.piece_len = (count + pieces - 1) / pieces;
    sp.n_pieces = sp.piece_len > 0 ? (count + sp.piece_len - 1) / sp.piece_len : 1;
    if (sp.n_pieces < 1) sp.n_pieces = 1;
#ifdef PT_PHASES   // the measurement build sums cycles in the control words: cleared per launch
    hipError_t e = hipMemsetAsync(d_sched, 0, sched_bytes(cam), stream);
    if (e != hipSuccess) return e;
#endif
    // (no memset: the previous launch left the words at zero -- retire_group)
    const long long items = (long long)sp.n_tiles * sp.n_pieces;
    const bool in_lds = sc.lds_nodes >= sc.n_nodes;
    // A scene of a handful of primitives keeps the launch of one workgroup of 4 waves per 4 items: its paths are a
    // segment or two, so what an item costs is set by WHICH tile it is (sky or sphere), the tickets race through
    // the cheap tiles, and the waves that draw the next pieces of the dear ones stand waiting for their predecessors
    // (cfg 2, 1920x1080 @ 256 spp: 12.5 ms this way, 17.5 ms with the persistent launch, 11 % of whose wave time is
    // that wait -- 11.0 against 14.7 ms since pieces are 64 samples; the headline frame waits 0.02 % of its time).
    if (!launch_is_persistent(sc)) {
        const dim3 grid((unsigned)((items + 3) / 4)), block(256);
        hipLaunchKernelGGL((path_kernel<false>), grid, block, group_lds_bytes(sc, PT_SMALL_N1_OFFSET), stream, sc, cam, rp, sp, d_radiance);
        return hipGetLastError();
    }
    if (sc.lds_nodes > PT_MAX_LDS_NODES) return hipErrorInvalidValue;   // pt_api.cpp stages at most that many
    size_t lds_bytes = group_lds_bytes(sc, PT_LDS_N1_OFFSET);
    if (sc.cand && !sc.ext && lds_bytes < 2 * (size_t)PT_LDS_N1_OFFSET) lds_bytes = 2 * (size_t)PT_LDS_N1_OFFSET;   // the waves' chains sit at the top of each half
    const long long groups = (items + WAVES_PER_GROUP - 1) / WAVES_PER_GROUP;
    const dim3 block(64 * WAVES_PER_GROUP);
    // as many workgroups as the chip holds at once; each wave draws items until none are left
    const PathArgs args{sc, cam, rp, sp, d_radiance};
    const void *fn = nullptr;
    if (!sc.ext) fn = in_lds ? (const void *)path_kernel_persistent<true, false> : (const void *)path_kernel_persistent<false, false>;
#ifndef PT_FAST_MODE
    // a small extension scene (the Cornell box: 35 nodes, 2 transforms) runs the instantiation whose LDS holds the
    // nodes in 8 KiB and, behind them, every lane's object-space rays of the walk it is on (pt_device.hpp obj_cache_fill)
    else if (PT_OBJ_CACHE && in_lds && sc.n_xforms <= PT_OBJ_MAX_XFORMS && sc.lds_xforms == sc.n_xforms &&
             sc.lds_nodes + 1 + 2 * sc.n_xforms <= PT_OBJ_N1_OFFSET / 16) {
        fn = (const void *)path_kernel_persistent<true, true, PT_OBJ_N1_OFFSET>;
        lds_bytes = PT_OBJ_LDS_BYTES;
    }
    else fn = in_lds ? (const void *)path_kernel_persistent<true, true> : (const void *)path_kernel_persistent<false, true>;
#else
    else return hipErrorInvalidValue;   // the FAST build carries the reference scenes' kernel only
#endif
    // How many groups the chip holds at once, and the LDS attribute of a launch beyond the default cap: asked once
    // per (device, kernel, LDS size) and kept, so that a launch is a memset and a kernel launch and nothing else --
    // which is also all a stream under HIP graph capture records.
    struct Resident { int dev; const void *fn; size_t lds; long long groups; };
    static std::mutex cache_lock;
    static Resident cache[32];
    static int cached = 0;
    int dev = 0;
    hipError_t q = hipGetDevice(&dev);
    if (q != hipSuccess) return q;
    long long resident = 0;
    {
        std::lock_guard<std::mutex> guard(cache_lock);
        for (int i = 0; i < cached; i++)
            if (cache[i].dev == dev && cache[i].fn == fn && cache[i].lds == lds_bytes) resident = cache[i].groups;
    }
    if (resident == 0) {
        int per_cu = 0, cus = 0;
        q = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (q != hipSuccess) return q;
        if (lds_bytes > 48 * 1024) {   // beyond the default cap on dynamic LDS per workgroup (a tree of more than 1023 nodes)
            q = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (q != hipSuccess) return q;
        }
        q = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * WAVES_PER_GROUP, lds_bytes);
        if (q != hipSuccess) return q;
        resident = (long long)per_cu * cus;
        if (resident < 1) resident = 1;
        std::lock_guard<std::mutex> guard(cache_lock);
        if (cached < 32) cache[cached++] = Resident{dev, fn, lds_bytes, resident};
    }
    const dim3 pgrid((unsigned)(groups < resident ? groups : resident));
    void *kargs[] = {(void *)&args};
    return hipLaunchKernel(fn, pgrid, block, kargs, lds_bytes, stream);
}

#ifndef PT_FAST_MODE

hipError_t launch_primary_hits(const Scene &sc, const Camera &cam, int S, int sample, int32_t *d_hit_id, float *d_hit_t,
                               hipStream_t stream)
{
    const int tiles = tiles_for(cam, 1);
    if (sc.ext) hipLaunchKernelGGL(primary_hit_kernel<true>, dim3(tiles), dim3(64), 0, stream, sc, cam, S, sample, d_hit_id, d_hit_t);
    else hipLaunchKernelGGL(primary_hit_kernel<false>, dim3(tiles), dim3(64), 0, stream, sc, cam, S, sample, d_hit_id, d_hit_t);
    return hipGetLastError();
}

hipError_t launch_resolve(const float *d_radiance, int n_pixels, int S, uint8_t *d_rgb8, hipStream_t stream)
{
    const int blocks = (n_pixels + 255) / 256;
    hipLaunchKernelGGL(resolve_kernel, dim3(blocks), dim3(256), 0, stream, d_radiance, n_pixels, (float)(S * S), d_rgb8);
    return hipGetLastError();
}

#endif   // !PT_FAST_MODE

}  // namespace PTK
