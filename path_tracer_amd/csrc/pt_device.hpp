// pt_device.hpp -- device-side arithmetic of the radiance hot path, gfx950.
//
// Every function states the reference lines it reproduces (tree at
// /root/reference).  The numeric contract is SURVEY.md section 9: each + - * /
// sqrt is ONE correctly rounded fp32 operation in the reference's order; the
// translation unit is compiled with -ffp-contract=off so hipcc never fuses
// a*b+c (Rust never does), and HIP's default correctly rounded fp32 divide and
// sqrt are kept.  Portable definitions P1-P3: RSQ(x) = 1/sqrt(x),
// fp64-evaluated atan2f/asinf, per-(pixel,sample) xorshift64 streams.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd {

struct f3 { float x, y, z; };

// ---- device scene -------------------------------------------------------------
struct Texture { const uint8_t *rgb8; int32_t width, height; };
struct Scene {
    const float4 *n0, *n1;   // one float4 each per node (layout below)
    // per primitive, 32 bytes = one half cache line for a hit's two loads; i = the primitive's RANK in the
    // reference's walk order (position of its Leaf in the tree's depth-first order), prim_ids[i] = the caller's id:
    //   prims[2 i]     = (cx, cy, cz, bits(kind | (texture index + 1) << 2))
    //   prims[2 i + 1] = its material's (albedo r, g, b, param); a Dielectric's: (schlick r0, 1 / ref_idx, -, ref_idx)
    const float4 *prims;
    const int32_t *prim_ids;
    const Texture *textures;
    int32_t n_nodes, n_spheres;   // n_nodes: slots of the node arrays
    int32_t end_slot;        // 16 * end_slot is the code that ends a walk: max(n_nodes, PT_FIRST_GLOBAL_SLOT), above every node's
                             // code and above the LDS offsets of the waves' chain entries, which are Split codes too
    int32_t root_code;       // node code of the root
    int32_t lds_nodes;       // n0[0 .. lds_nodes) is staged in LDS by the path kernel
    int32_t bounded;         // every sphere has |c_i| <= 2^32 and |r| <= 2^32 (see regular_ray)
    // extension scenes only (pt_upload_scene_ext)
    const float4 *xforms;    // per transform: (offset.xyz, sin theta), (cos theta, -, -, -)
    int32_t ext;             // 1: rendered by the extension's kernel instantiation
    int32_t lds_xforms;      // the path kernel stages this many transforms (all of them, or 0) in LDS behind the nodes
    int32_t n_xforms;        // transforms of the scene (object-ray cache of the extension's kernel: PT_OBJ_MAX_XFORMS at most)
    // Primary-ray candidate lists (pt_kernels.hip primary_list_kernel), or NULL: per 8x8 pixel block of the current
    // camera's frame, PT_PRIMARY_K leaf codes -- every sphere a camera ray through the block can possibly hit --
    // ended by 0; a first entry of 1 = too many: walk the tree.  cand_tiles_x = blocks per row.
    const int32_t *cand;
    int32_t cand_tiles_x;
    const int32_t *parent_codes;   // per primitive rank: the code of its Leaf's parent Split
};
struct Camera {               // camera.rs:4-11
    f3 llc, origin, horizontal, vertical;
    int32_t width, height;
};

#define PTD __device__ __forceinline__

// Everything below is arithmetic.  It exists in two builds that must never mix: the EXACT one (this
// header as the library's translation units see it: every operation rounded once, in the reference's
// order) and the FAST one (pt_kernels_fast.hip: contraction on, hardware reciprocal / square root
// estimates, opt-in through pt_set_mode) -- hence the inline namespace, which keeps the two sets of
// inline functions apart while the types above, which cross the launch interface, stay shared.
#ifdef PT_FAST_MODE
inline namespace fastmath {
#else
inline namespace exactmath {
#endif

PTD f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PTD f3 fill(float v) { return mk(v, v, v); }                                     // vec.rs:79
PTD f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }     // vec.rs:145
PTD f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }     // vec.rs:153
PTD f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }     // vec.rs:177
PTD f3 operator/(f3 a, f3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }     // vec.rs:185
PTD f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }        // vec.rs:161 (splat)
PTD f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }        // vec.rs:169 (splat)
PTD f3 neg(f3 a) { return a * -1.0f; }                                           // vec.rs:137-142
PTD float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }        // vec.rs:32-36,99-101
// the same with contraction off whatever the build: three products and two sums, each rounded once
PTD float dot_unfused(f3 a, f3 b)
{
#pragma clang fp contract(off)
    return (a.x * b.x + a.y * b.y) + a.z * b.z;
}
// Short forms of the three unary IEEE operations on the path.  Each returns, bit for bit, what
// the correctly rounded operation returns -- for EVERY fp32 input: inside the guard range one
// hardware estimate (v_rcp_f32 / v_rsq_f32, 1 ulp) plus one fused correction step rounds
// correctly (checked exhaustively over all 2^32 inputs on the GPU: tools/verify/unary_exhaustive.hip,
// tests/test_gpu_parity.py::test_unary_short_forms); outside it (zero, denormal, huge, inf, NaN,
// negative) the full expansion runs.  6-10 instructions instead of 12-29.  The fma here is
// explicit and internal to one operation; no a*b+c of the reference is ever contracted.
PTD float rcp_core(float s)
{
    float r = __builtin_amdgcn_rcpf(s);
    float e = __builtin_fmaf(-s, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
PTD float sqrt_core(float x)
{
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
// 1.0f / x; guard |x| in [2^-95, 2^95]
PTD float x_rcp(float x)
{
    if (__builtin_expect((__float_as_uint(x) & 0x7fffffffu) - 0x10000000u <= 0x5f000000u, 1)) return rcp_core(x);
    return 1.0f / x;
}
// sqrtf(x); guard x in [2^-95, FLT_MAX]
PTD float x_sqrt(float x)
{
    if (__builtin_expect(__float_as_uint(x) - 0x10000000u <= 0x6f7fffffu, 1)) return sqrt_core(x);
    return __builtin_sqrtf(x);
}
// P1: portable RSQ in place of _mm_rsqrt_ps (vec.rs:38-44): 1.0f / sqrtf(x), two roundings
PTD float x_rsq(float x)
{
    if (__builtin_expect(__float_as_uint(x) - 0x10000000u <= 0x6f7fffffu, 1)) return rcp_core(sqrt_core(x));
    return 1.0f / __builtin_sqrtf(x);
}
#ifdef PT_FAST_MODE
#ifndef PT_FAST_LEAF
// Sphere::trace in the FAST build: 0 = exactly as in the EXACT build; 1 = b, c and the discriminant un-fused (they
// cancel: their rounding decides grazing hits), roots through the sqrt estimate and * 1/a; 2 = everything fused.
// Measured on the RTOW scene at 4096 spp, pixels within 1/255 of EXACT: 99.994 % / 99.995 % / 99.78 % -- with 2
// the horizon of the r = 1000 ground sphere and every silhouette differ systematically, whatever the spp
// (tools/fast_diag.py, tools/fast_where.py); 1 costs 2 % over 2.
#define PT_FAST_LEAF 1
#endif
// FAST build (pt_set_mode(PT_FAST)): the hardware estimates as they are (1 ulp), no correction step
PTD float p_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
PTD float p_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
PTD float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
#if PT_FAST_LEAF == 0
PTD float walk_rcp(float a) { return rcp_core(a); }
PTD float leaf_sqrt(float x) { return x_sqrt(x); }
#else
PTD float walk_rcp(float a) { return __builtin_amdgcn_rcpf(a); }
PTD float leaf_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#endif
#else
PTD float walk_rcp(float a) { return rcp_core(a); }   // RN(1 / dot(d, d)) for REGULAR rays (regular_ray)
PTD float leaf_sqrt(float x) { return x_sqrt(x); }
PTD float p_rcp(float x) { return x_rcp(x); }
PTD float p_sqrt(float x) { return x_sqrt(x); }
PTD float rsq(float x) { return x_rsq(x); }
#endif
PTD f3 normalized(f3 a) { return a * rsq(dot(a, a)); }                           // vec.rs:95-97

// _mm_min_ps / _mm_max_ps lane semantics (vec.rs:50-51): second operand if either is NaN
PTD float sse_min(float a, float b) { return a < b ? a : b; }
PTD float sse_max(float a, float b) { return a > b ? a : b; }

// Rust `as i32` / `as u8`: truncate, saturate, NaN -> 0 (picture.rs:21-22, raytracer.rs:121)
PTD int32_t trunc_sat_i32(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
PTD uint8_t trunc_sat_u8(float f)
{
    if (f != f) return 0;
    if (f >= 255.0f) return 255;
    if (f <= 0.0f) return 0;
    return (uint8_t)f;
}

// ---- RNG (rng.rs) -----------------------------------------------------------
PTD uint64_t rng_raw(uint64_t &s)                                                // rng.rs:50-60
{
    uint64_t x = s;
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    s = x;
    return x;
}
PTD float rng_f32(uint64_t &s)                                                   // rng.rs:76-80
{
    uint32_t v = (uint32_t)rng_raw(s);
    return (float)(v >> 8) / 16777216.0f;
}
PTD float rng_range(uint64_t &s, float mn, float mx) { return rng_f32(s) * (mx - mn) + mn; } // rng.rs:83-85
// P3: state of the (pixel, sample) stream after Rng::with_seed (rng.rs:27-35): the stream is
// with_seed(W^-1(z)) for W = with_seed's ten rounds, a bijection, so the rounds are not run here
PTD uint64_t path_rng(uint64_t seed, uint32_t pixel, uint32_t sample)
{
    uint64_t k = ((uint64_t)pixel << 32) | (uint64_t)sample;
    uint64_t z = seed + (k + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    if (z == 0) z = 0x9E3779B97F4A7C15ull;
    return z;
}

// ---- math/mod.rs ------------------------------------------------------------
PTD f3 random_in_unit_sphere(uint64_t &rng)                                      // math/mod.rs:13-19
{
    float x = rng_range(rng, -1.0f, 1.0f);
    float y = rng_range(rng, -1.0f, 1.0f);
    float z = rng_range(rng, -1.0f, 1.0f);
    return normalized(mk(x, y, z));
}
PTD f3 reflect(f3 d, f3 n) { return d - ((n * dot(d, n)) * 2.0f); }              // math/mod.rs:21-23
PTD bool refract(f3 d, f3 n, float k, f3 &out)                                   // math/mod.rs:25-34
{
    float dt = dot(d, n);
    float D = 1.0f - (k * k) * (1.0f - dt * dt);
    if (D > 0.0f) {
        out = ((d - n * dt) * k) - (n * p_sqrt(D));
        return true;
    }
    return false;
}
// r0 of math/mod.rs:37-38, a function of the material alone: computed once at upload (same fp32 ops)
inline __host__ __device__ float schlick_r0(float ref_idx)
{
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    return r0 * r0;
}
PTD float schlick(float cosine, float r0)                                        // math/mod.rs:36-45 (v^5 as written), from r0 on
{
    float v = r0 + (1.0f - r0) * (1.0f - cosine);
    float v2 = v * v;
    float v4 = v2 * v2;
    return v4 * v;
}

// ---- P2: portable atan2f / asinf (sphere.rs:9-10), fp64 core -----------------
PTD double atan_unit(double q)
{
    // pi / 4 handed over in scalar registers: as a 64-bit vector literal the compiler hoists it out of the path loop
    // and, with no vector register to spare, spills it -- a store per lane at the start of every item
    int qp_hi = 0x3fe921fb, qp_lo = 0x54442d18;
    asm("" : "+s"(qp_hi), "+s"(qp_lo));
    const double quarter_pi = __hiloint2double(qp_hi, qp_lo);
    double base = 0.0, t = q;
    if (q > 0.41421356237309504880) { t = (q - 1.0) / (q + 1.0); base = quarter_pi; }   // 0.78539816339744830962
    double z = t * t;
    double p = 1.0 / 45.0;
#pragma unroll
    for (int k = 21; k >= 0; k--) {
        double c = 1.0 / (double)(2 * k + 1);
        p = c - z * p;
    }
    return base + t * p;
}
PTD double atan2_core(double y, double x)
{
    if (y != y || x != x) return __builtin_nan("");
    double ay = __builtin_fabs(y), ax = __builtin_fabs(x), a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ay > ax) a = 1.57079632679489661923 - atan_unit(ax / ay);
    else a = atan_unit(ay / ax);
    if (__builtin_signbit(x)) a = 3.14159265358979323846 - a;
    return __builtin_signbit(y) ? -a : a;
}
PTD float p_atan2f(float y, float x) { return (float)atan2_core((double)y, (double)x); }
PTD float p_asinf(float y)
{
    double yd = (double)y;
    if (!(__builtin_fabs(yd) <= 1.0)) return __builtin_nanf("");
    double c = __builtin_sqrt((1.0 - yd) * (1.0 + yd));
    return (float)atan2_core(yd, c);
}

struct Ray { f3 o, d; };

PTD Ray camera_ray(const Camera &c, float u, float v)                            // camera.rs:37-44
{
#pragma clang fp contract(off)   // also in the FAST build: primary rays are the reference's, bit for bit
    Ray r;
    r.o = c.origin;
    r.d = ((c.llc + c.horizontal * u) + c.vertical * v) - c.origin;
    return r;
}
// sample s = sx*S + sy of the grid of raytracer.rs:77-86
PTD Ray sample_ray(const Camera &c, int S, int px, int py, int s)
{
    int sx = s / S, sy = s - sx * S;
    float x = (float)px + ((float)sx / (float)(S - 1));
    float u = x / (float)c.width;
    float y = (float)py + ((float)sy / (float)(S - 1));
    float v = 1.0f - (y / (float)c.height);
    return camera_ray(c, u, v);
}

// x / a from ra = RN(1/a) (correctly rounded reciprocal): a product and two fused residual
// corrections instead of the 12-instruction IEEE expansion.  Correctly rounded for every x, a
// whose quotient and residuals stay normal:
//   q0 = RN(x*ra) is within 2 ulp of x/a;  q1 = RN(q0 + r0*ra) is faithful (one of the two floats
//   around x/a), so r1 = x - a*q1 is exact;  q2 = RN(q1 + r1*ra) is then RN(x/a) by Markstein's
//   division theorem (1990; Cornea/Harrison/Tang, "Scientific Computing on Itanium", Thm 8.3), whose
//   only hypotheses are "q1 faithful" and "ra = RN(1/a)" -- it holds for EVERY mantissa of a,
//   including the all-ones one (2^24 - 1): that mantissa is the exception of the theorem that
//   REFINES A RECIPROCAL, which is not used here -- ra comes from rcp_core, whose result is checked
//   against 1/a for all 2^32 inputs (pt_selftest_unary).
// Two users: the sample grid of trace_pixel, whose divisors are S-1, W and H (pt_selftest_grid
// compares every grid coordinate of a frame size with the IEEE divisions), and Sphere::trace's
// divisions by a = dot(d, d) for REGULAR rays (regular_ray below), where a is about 1 - 2^-24 --
// the all-ones mantissa -- for most normalised directions: pt_selftest_division pins that mantissa
// in a third of its pairs (mode 2), over all exponents of the class (tests/test_gpu_parity.py::test_reciprocal_division).
PTD float div_by(float x, float a, float ra)
{
    float q = x * ra;
    float r = __builtin_fmaf(-a, q, x);
    q = __builtin_fmaf(r, ra, q);
    r = __builtin_fmaf(-a, q, x);
    return __builtin_fmaf(r, ra, q);
}
// sample (sx, sy) of the grid of raytracer.rs:77-86 with the three divisions through div_by
PTD Ray sample_ray_grid(const Camera &c, int S, float inv_s1, float inv_w, float inv_h, int px, int py, int sx, int sy)
{
    const float s1 = (float)(S - 1);
    float x = (float)px + div_by((float)sx, s1, inv_s1);
    float u = div_by(x, (float)c.width, inv_w);
    float y = (float)py + div_by((float)sy, s1, inv_s1);
    float v = 1.0f - div_by(y, (float)c.height, inv_h);
    return camera_ray(c, u, v);
}

PTD f3 sphere_uv_color(const Scene &sc, int tex, f3 n)                           // sphere.rs:6-18 + picture.rs:17-38
{
    float phi = p_atan2f(n.z, n.x);
    float theta = p_asinf(n.y);
    const float pi = 3.14159265358979323846f;
    float u = 1.0f - (phi + pi) / (2.0f * pi);
    float v = (theta + pi / 2.0f) / pi;
    Texture t = sc.textures[tex];
    int32_t i = trunc_sat_i32(u * (float)t.width);
    int32_t j = trunc_sat_i32(((1.0f - v) * (float)t.height) - 0.001f);
    i = i < t.width - 1 ? i : t.width - 1;   i = i > 0 ? i : 0;                  // picture.rs:24-26
    j = j < t.height - 1 ? j : t.height - 1; j = j > 0 ? j : 0;
    const uint8_t *px = t.rgb8 + 3 * ((size_t)j * (size_t)t.width + (size_t)i);
    return mk((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
}

// ---- Scene::trace (scene/mod.rs:24-34) over the flattened BvhNode tree ---------
//
// A node CODE is the byte offset (16 * slot) of the node in the arrays n0[], n1[] with the
// sign bit set when the node is a Leaf; the code 16 * n_nodes (no flag) ends the walk.  Slots are
// numbered breadth-first (top of the tree first); the walk does not depend on
// the numbering, it follows the codes stored in the nodes:
//          n0[slot]                                   n1[slot]
//   Split: (bmin.x, bmin.y, bmin.z, bits(skip code))  (bmax.x, bmax.y, bmax.z, bits(next code))
//   Leaf : (c.x,    c.y,    c.z,    r*r)              (-, -, bits(prim), bits(next code))
// next = the following node in DFS pre-order, skip = the first node after this
// node's subtree, so the kind of the NEXT node is known before its data arrives.  The walk is the
// iterative form of BvhNode::trace (bvh.rs:106-127): DFS pre-order, left before
// right, max_t = closest hit so far; a Split is entered only if its own box
// passes AABB::intersect (aabb.rs:25-36); a Leaf is traced without a box test
// (bvh.rs:109); strict '<' keeps the first hit in DFS order on ties (sphere.rs:57,63).
//
// REGULAR = every component of origin and of 1/direction is finite.  Then no
// operand of the slab test can be NaN and IEEE min/max (one v_min/v_max each)
// give the same tmin/tmax as the reference's _mm_min_ps/_mm_max_ps; the general
// variant reproduces the SSE "second operand on NaN" rule explicitly.
constexpr float T_MIN = 0.001f;                                                  // scene/mod.rs:25
constexpr float T_MAX = 3.402823466e+38f;                                        // f32::MAX, scene/mod.rs:27
constexpr int32_t LEAF_BIT = (int32_t)0x80000000;

struct Hit { float t; int32_t prim; };   // prim: device index = rank in the reference's walk order, -1 = none

// The acceptance test of Sphere::trace (sphere.rs:57,63: t < max_t && t > min_t) for a root of the primitive
// of rank `rank`, given the closest hit so far.  The reference walks the primitives in rank order and keeps
// the first of two equal roots (strict '<'); a walk in another order (PT_WALK_VIEW) reaches the same choice by
// letting the lower rank win a tie.  In rank order the tie clause never fires: ranks only grow along the walk.
PTD bool closer(float sol, int32_t rank, const Hit &h)
{
    return (sol < h.t || (sol == h.t && rank < h.prim)) && sol > T_MIN;
}

// codes count in units of 2^-PT_CODE_SHIFT slots: 4 = byte offsets (no shift per node visit)
#ifndef PT_CODE_SHIFT
#define PT_CODE_SHIFT 4
#endif
PTD int code_index(int32_t code) { return (code & 0x7fffffff) >> PT_CODE_SHIFT; }
PTD int code_offset(int32_t code) { return (code & 0x7fffffff) << (4 - PT_CODE_SHIFT); }

// One-instruction IEEE min/max.  Only used where no operand can be NaN (REGULAR
// rays), where they equal both _mm_min_ps/_mm_max_ps and Rust's f32::min/max;
// the builtins would add a canonicalising v_max(x, x) per operand.
PTD float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
PTD float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
PTD float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
PTD float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// one v_med3_f32 (no canonicalising v_max; constants come from SGPRs)
PTD float vmed3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// AABB::intersect (aabb.rs:25-36) for the Split whose data is (n0, n1); returns the next code.
// REGULAR (no NaN anywhere): with lo_k = min(v0.k, v1.k), hi_k = max(v0.k, v1.k) the test is
//   max(T_MIN, lo_x, lo_y, lo_z) < min(closest, hi_x, hi_y, hi_z).
// The chains e = med3(med3(med3(T_MIN, x), y), z), f = med3(med3(med3(closest, x), y), z)
// (med3(e, v0.k, v1.k) = clamp e into [lo_k, hi_k]) give the same e < f DECISION for every
// ordering of the eight values -- checked exhaustively over all weak orders of 8 elements
// (tests/test_host_logic.py::test_med3_slab_identity) -- with 6 half-rate VALU ops instead of 10.
template <bool REGULAR>
PTD int32_t split_step(const float4 n0, const float4 n1, const f3 o, const f3 inv, const float closest)
{
    f3 v0 = (mk(n0.x, n0.y, n0.z) - o) * inv;
    f3 v1 = (mk(n1.x, n1.y, n1.z) - o) * inv;
    float tmin, tmax;
    if (REGULAR) {
        tmin = vmed3(vmed3(vmed3(T_MIN, v0.x, v1.x), v0.y, v1.y), v0.z, v1.z);
        tmax = vmed3(vmed3(vmed3(closest, v0.x, v1.x), v0.y, v1.y), v0.z, v1.z);
    } else {
        float mnx = sse_min(v0.x, v1.x), mny = sse_min(v0.y, v1.y), mnz = sse_min(v0.z, v1.z);
        float mxx = sse_max(v0.x, v1.x), mxy = sse_max(v0.y, v1.y), mxz = sse_max(v0.z, v1.z);
        tmin = __builtin_fmaxf(T_MIN, __builtin_fmaxf(mnx, __builtin_fmaxf(mny, mnz)));
        tmax = __builtin_fminf(closest, __builtin_fminf(mxx, __builtin_fminf(mxy, mxz)));
    }
    return (tmin < tmax) ? __float_as_int(n1.w) : __float_as_int(n0.w);
}

#ifdef PT_FAST_MODE
// FAST build: the slab test with one fused multiply-add per plane, t = b * inv + (-(o * inv)), the product
// o * inv taken once per walk (noi).  Not the reference's rounding: results are within the north-star
// tolerance of it, not bit-equal (pt_set_mode).
PTD int32_t split_step_fast(const float4 n0, const float4 n1, const f3 inv, const f3 noi, const float closest)
{
    const f3 v0 = mk(__builtin_fmaf(n0.x, inv.x, noi.x), __builtin_fmaf(n0.y, inv.y, noi.y), __builtin_fmaf(n0.z, inv.z, noi.z));
    const f3 v1 = mk(__builtin_fmaf(n1.x, inv.x, noi.x), __builtin_fmaf(n1.y, inv.y, noi.y), __builtin_fmaf(n1.z, inv.z, noi.z));
    const float tmin = vmed3(vmed3(vmed3(T_MIN, v0.x, v1.x), v0.y, v1.y), v0.z, v1.z);
    const float tmax = vmed3(vmed3(vmed3(closest, v0.x, v1.x), v0.y, v1.y), v0.z, v1.z);
    return (tmin < tmax) ? __float_as_int(n1.w) : __float_as_int(n0.w);
}
#endif

// Sphere::trace (sphere.rs:46-69) for the Leaf whose data is (n0, n1); `a` = dot(d, d).
// FAST (wave-uniform; REGULAR rays only, see regular_ray): the divisions by a = dot(d, d) go
// through div_by with ra = RN(1/a), computed once per walk.
//
// In two parts, because the two halves of a Leaf arrive at different times (n0 from LDS, n1 -- rank and next
// code -- through the L1, several hundred cycles later under load: a Leaf step that reads n1 before its
// arithmetic is done stalls on it, +14 % on the whole kernel):
//   leaf_candidate  everything that needs n0 only: the root this sphere offers against the closest hit so
//                   far, and whether the reference's test accepts it outright (`sure`) or it exactly TIES
//                   the closest hit (`maybe`);
//   leaf_accept     with n1: a tie goes to the lower rank (see closer()), then the hit is updated.
// sphere.rs:55-66 with closer()'s tie rule, case by case (h.t > T_MIN always, so a tie is in range):
//   first root r1:  r1 < h.t and r1 > T_MIN -> sure;  r1 == h.t -> maybe (then the second root, r2 >= r1 = h.t,
//                   can only tie as well, with the same rank comparison: nothing more to try);
//   otherwise r2:   r2 < h.t and r2 > T_MIN -> sure;  r2 == h.t -> maybe.
// Written flat: the two roots are computed under their conditions, every comparison is made where all the
// lanes of the step are active again, so that each flag is a lane mask in scalar registers, combined by scalar
// instructions (a flag that crosses the end of a branch becomes a 0/1 vector register instead, with vector
// instructions to pack and unpack it: +3.5 % on the whole kernel).  The tie rule costs three compares per Leaf
// step: two `==` here, the rank comparison in leaf_accept.
struct LeafCandidate { float t; bool sure, tie; };
PTD LeafCandidate leaf_candidate(const float4 n0, const f3 o, const f3 d, const float a, const float ra, const bool FAST, const float closest)
{
#if defined(PT_FAST_MODE) && PT_FAST_LEAF < 2
#pragma clang fp contract(off)   // the discriminant cancels: its rounding decides grazing hits (horizon, silhouettes)
#endif
    f3 oc = o - mk(n0.x, n0.y, n0.z);
#if defined(PT_FAST_MODE) && PT_FAST_LEAF < 2
    float b = dot_unfused(oc, d);
    float c = dot_unfused(oc, oc) - n0.w;
#else
    float b = dot(oc, d);
    float c = dot(oc, oc) - n0.w;
#endif
    const float disc = b * b - a * c;
    const bool hit = disc > 0.0f;
    float sq = 0.0f, sol1 = 0.0f, sol2 = 0.0f;
    if (hit) {
        sq = leaf_sqrt(disc);
#if defined(PT_FAST_MODE) && PT_FAST_LEAF >= 1
        (void)FAST;
        sol1 = (-b - sq) * ra;
#else
        sol1 = FAST ? div_by(-b - sq, a, ra) : (-b - sq) / a;
#endif
    }
    const bool first = (bool)((int)hit & (int)(sol1 < closest) & (int)(sol1 > T_MIN));
    const bool need2 = (bool)((int)hit & (int)!first);
    if (need2) {
#if defined(PT_FAST_MODE) && PT_FAST_LEAF >= 1
        sol2 = (-b + sq) * ra;
#else
        sol2 = FAST ? div_by(-b + sq, a, ra) : (-b + sq) / a;
#endif
    }
    LeafCandidate k;
    k.sure = (bool)((int)first | ((int)need2 & (int)(sol2 < closest) & (int)(sol2 > T_MIN)));
#ifndef PT_NO_TIE   // (PT_NO_TIE: the FAST build and measurement builds -- ties go to the walk's first candidate)
    // a first root that ties is followed by a second one >= it, which can only tie as well: either way the lane
    // offers `closest` itself (h.t stays as it is) and the rank decides
    k.tie = (bool)((int)need2 & ((int)(sol1 == closest) | (int)(sol2 == closest)));
#else
    k.tie = false;
#endif
    k.t = first ? sol1 : sol2;
    return k;
}
PTD void leaf_accept(const LeafCandidate k, const float4 n1, Hit &h)
{
    const int32_t rank = __float_as_int(n1.z);
    const bool take = (bool)((int)k.sure | ((int)k.tie & (int)(rank < h.prim)));
    h.t = k.sure ? k.t : h.t;
    h.prim = take ? rank : h.prim;
}
PTD void leaf_step(const float4 n0, const float4 n1, const f3 o, const f3 d, const float a, const float ra, const bool FAST, Hit &h)
{
    leaf_accept(leaf_candidate(n0, o, d, a, ra, FAST, h.t), n1, h);
}

// ---- extension: rects and Translate(RotateY(..)) instances (NOT in the reference) ----------
// Leaf of a rect:  n0 = (a0, a1, b0, b1)   n1 = (k, bits(flags), bits(prim), bits(next code))
// flags: bit 0 = rect, bits 1-2 = axis, bit 3 = flip, bits 4.. = transform index + 1; a sphere's Leaf has flags 0.
struct Xform { f3 off; float s, c; };
// xf_lds: LDS byte address of the staged transform table (path kernel), or XF_GLOBAL: read sc.xforms.  A
// vector-memory load costs the CU's address unit as much as a node visit does; an LDS read
// is free next to it, and a rect test of an instance needs its transform every time.
constexpr uint32_t XF_GLOBAL = 0xffffffffu;
PTD float4 lds_f4(uint32_t addr)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const v4f lds_v4f;
    const v4f t = *(lds_v4f *)(uintptr_t)addr;
    return make_float4(t.x, t.y, t.z, t.w);
}
PTD Xform load_xform(const Scene &sc, int idx, const uint32_t xf_lds)
{
    float4 a, b;
    if (xf_lds != XF_GLOBAL) { a = lds_f4(xf_lds + 32u * (uint32_t)idx); b = lds_f4(xf_lds + 32u * (uint32_t)idx + 16u); }
    else { a = sc.xforms[2 * idx]; b = sc.xforms[2 * idx + 1]; }
    Xform x; x.off = mk(a.x, a.y, a.z); x.s = a.w; x.c = b.x;
    return x;
}
PTD f3 rot_to_object(const Xform &x, f3 p) { return mk(x.c * p.x - x.s * p.z, p.y, x.s * p.x + x.c * p.z); }
PTD f3 rot_to_world(const Xform &x, f3 p) { return mk(x.c * p.x + x.s * p.z, p.y, x.c * p.z - x.s * p.x); }
PTD float pick(f3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }
// the ray in the rect's object space (un-normalised: t is the same in both spaces)
PTD void rect_object_ray(const Scene &sc, int flags, f3 &o, f3 &d, const uint32_t xf_lds)
{
    const int xf = (flags >> 4) - 1;
    if (xf >= 0) {
        const Xform x = load_xform(sc, xf, xf_lds);
        o = rot_to_object(x, o - x.off);
        d = rot_to_object(x, d);
    }
}
// The object-ray cache of the extension's path kernel (pt_kernels.hip, N1OFF == PT_OBJ_N1_OFFSET): a ray's object-space
// form depends on (ray, transform) only, and a Cornell-box walk makes seven rect tests, two thirds of them on the faces
// of the two instanced boxes -- each of which transformed the ray again (15 operations) and then picked three of its
// six components by the rect's axis (8 selects).  At the start of a walk the lane stores the ray as it is (slot 0)
// and in the space of every transform (slots 1 .., the SAME operations in the same order: same bits) in LDS, one
// (origin, direction) pair per component and lane; a rect test reads the three pairs its axis asks for.
//   layout: slot s, component k: PT_OBJ_BASE + (3 s + k) * PT_OBJ_PLANE + 8 * thread   (float2 = (o[k], d[k]))
#define PT_OBJ_N1_OFFSET 4096            // n1[] of the staged nodes, the ticket and the transforms: 256 entries in all
#define PT_OBJ_BASE 8192
#define PT_OBJ_PLANE 8192                // 1024 threads x 8 bytes
#define PT_OBJ_MAX_XFORMS 2
#define PT_OBJ_LDS_BYTES (PT_OBJ_BASE + 3 * (1 + PT_OBJ_MAX_XFORMS) * PT_OBJ_PLANE)   // 80 KiB: two groups per CU
PTD float2 lds_f2(uint32_t addr)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const v2f lds_v2f;
    const v2f t = *(lds_v2f *)(uintptr_t)addr;
    return make_float2(t.x, t.y);
}
PTD void lds_put_f2(uint32_t addr, float a, float b)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) v2f lds_v2f;
    const v2f t = {a, b};
    *(lds_v2f *)(uintptr_t)addr = t;
}
// `at`: PT_OBJ_BASE + 8 * thread.  Slot 0 = the ray itself, slot 1 + i = in the space of transform i.
PTD void obj_cache_fill(const Scene &sc, const uint32_t at, const f3 o, const f3 d, const uint32_t xf_lds)
{
    lds_put_f2(at, o.x, d.x); lds_put_f2(at + PT_OBJ_PLANE, o.y, d.y); lds_put_f2(at + 2 * PT_OBJ_PLANE, o.z, d.z);
    for (int i = 0; i < sc.n_xforms; i++) {                                      // wave-uniform, PT_OBJ_MAX_XFORMS at most
        const Xform x = load_xform(sc, i, xf_lds);
        const f3 oo = rot_to_object(x, o - x.off), od = rot_to_object(x, d);     // rect_object_ray's operations
        const uint32_t s = at + (uint32_t)(3 * (i + 1)) * PT_OBJ_PLANE;
        lds_put_f2(s, oo.x, od.x); lds_put_f2(s + PT_OBJ_PLANE, oo.y, od.y); lds_put_f2(s + 2 * PT_OBJ_PLANE, oo.z, od.z);
    }
}
// rect_step with the ray's components read from the cache (same operations on the same values as rect_step)
PTD void rect_step_cached(const float4 n0, const float4 n1, const uint32_t at, Hit &h)
{
    const int flags = __float_as_int(n1.y);
    const int axis = (flags >> 1) & 3;
    const uint32_t s = __umul24((uint32_t)(flags >> 4), 3u * PT_OBJ_PLANE) + at;  // bits 4..: transform index + 1 = the slot
    const uint32_t ia = axis == 0 ? PT_OBJ_PLANE : 0u, ib = axis == 2 ? PT_OBJ_PLANE : 2u * PT_OBJ_PLANE;
    // the three reads go out together, ahead of the division (one LDS round trip per test, not two)
    float2 k = lds_f2(s + (uint32_t)axis * PT_OBJ_PLANE), pa = lds_f2(s + ia), pb = lds_f2(s + ib);
    asm volatile("" : "+v"(k.x), "+v"(k.y), "+v"(pa.x), "+v"(pa.y), "+v"(pb.x), "+v"(pb.y));
    const float t = (n1.x - k.x) / k.y;
    if (!(t < h.t && t > T_MIN)) return;
    const float a = pa.x + pa.y * t, b = pb.x + pb.y * t;
    if (!(a >= n0.x && a <= n0.y && b >= n0.z && b <= n0.w)) return;
    h.t = t; h.prim = __float_as_int(n1.z);
}
// Traceable::trace of a rect: t = (k - o[axis]) / d[axis], accepted iff t < max_t && t > min_t (the
// comparison of sphere.rs:57) and the point lies in [a0, a1] x [b0, b1]; a NaN fails every test
PTD void rect_step(const Scene &sc, const float4 n0, const float4 n1, f3 o, f3 d, Hit &h, const uint32_t xf_lds = XF_GLOBAL)
{
    const int flags = __float_as_int(n1.y);
    const int axis = (flags >> 1) & 3;
    rect_object_ray(sc, flags, o, d, xf_lds);
    const int ia = axis == 0 ? 1 : 0, ib = axis == 2 ? 1 : 2;
    const float t = (n1.x - pick(o, axis)) / pick(d, axis);
    if (!(t < h.t && t > T_MIN)) return;
    const float a = pick(o, ia) + pick(d, ia) * t, b = pick(o, ib) + pick(d, ib) * t;
    if (!(a >= n0.x && a <= n0.y && b >= n0.z && b <= n0.w)) return;
    h.t = t; h.prim = __float_as_int(n1.z);
}

PTD bool finite3(f3 v)
{
    return __builtin_isfinite(v.x) && __builtin_isfinite(v.y) && __builtin_isfinite(v.z);
}

// REGULAR ray: the class of rays the short exact forms are proven for.
//  * 1/d finite and o finite: the slab test has no NaN, so the med3 chain of split_step<true>
//    equals the reference's min/max chain.
//  * |o_i| < 2^32, a = d.d in [2^-32, 2^32], and every sphere of the scene has |c_i|, |r| <= 2^32
//    (`bounded`, checked at upload): then in Sphere::trace |b| < 2^51, |disc| < 2^103,
//    |-b +- sqrt(disc)| < 2^53, nothing overflows, and a root that can be accepted (t > T_MIN)
//    has every intermediate of div_by far inside the normal range, where the two fused
//    corrections return the correctly rounded quotient (Markstein; tools/verify/div_recip.hip).
//    A root below 2^-20 comes out below T_MIN from both forms and is rejected by both.
PTD bool regular_ray(const f3 inv, const f3 o, const float a, const bool bounded)
{
    const float lim = 4294967296.0f;   // 2^32
    const bool o_ok = __builtin_fabsf(o.x) < lim && __builtin_fabsf(o.y) < lim && __builtin_fabsf(o.z) < lim;
    const bool a_ok = __float_as_uint(a) - 0x2f800000u <= 0x20000000u;
    return finite3(inv) && o_ok && a_ok && bounded;
}

// whole-ray walk (used by the primary-hit kernel; the path kernel runs the same
// steps from its state machine)
template <bool EXT = false>
PTD Hit scene_trace(const Scene &sc, const Ray &ray)
{
    const f3 inv = mk(p_rcp(ray.d.x), p_rcp(ray.d.y), p_rcp(ray.d.z));           // scene/mod.rs:31: (1,1,1) / dir
    const float a = dot(ray.d, ray.d);                                           // sphere.rs:49
    const bool regular = regular_ray(inv, ray.o, a, sc.bounded != 0);
    const float ra = rcp_core(a);                                                // read by REGULAR rays only
    Hit h; h.t = T_MAX; h.prim = -1;
    int32_t code = sc.root_code;
    while (code_index(code) < sc.end_slot) {   // code_index(end code) == end_slot
        const int i = code_index(code);
        const float4 n0 = sc.n0[i], n1 = sc.n1[i];
        if (code < 0) {
            if (EXT && (__float_as_int(n1.y) & 1)) rect_step(sc, n0, n1, ray.o, ray.d, h);
            else leaf_step(n0, n1, ray.o, ray.d, a, ra, regular, h);
            code = __float_as_int(n1.w);
        }
        else code = regular ? split_step<true>(n0, n1, ray.o, inv, h.t) : split_step<false>(n0, n1, ray.o, inv, h.t);
    }
    return h;
}

// ---- Material::scatter (material/*.rs) -----------------------------------------
// The three Material::scatter bodies for a hit at p with normal n; returns false when the path is
// absorbed (metal.rs:29-31).  They share their expensive pieces, so those are written once
// and run once per step for the lanes of every material that needs them (each lane still
// executes exactly its material's operations, in its material's order):
//   reflect(d, n)            metal.rs:23, dielectric.rs:32
//   random_in_unit_sphere    lambertian.rs:33, metal.rs:26 (drawn only if the ray is not absorbed)
//   Ray::new's normalisation ray.rs:19-24
// kind 3 (EXT only: a diffuse light) scatters nothing and draws nothing.
template <bool EXT>
PTD bool scatter_at(const Scene &sc, const Ray &ray, const f3 p, const f3 n, const int kind, const int tex, const float4 ma,
                    uint64_t &rng, f3 &att, Ray &out)
{
    const f3 reflected = reflect(ray.d, n);                                      // unused by Lambertian lanes
    bool alive = !EXT || kind != 3;
    if (kind == 1) alive = dot(reflected, n) > 0.0f;                             // metal.rs:25, else None (:29-31)
    f3 unit = fill(0.0f);
    if (kind == 0 || (kind == 1 && alive)) unit = random_in_unit_sphere(rng);
    f3 nd;
    if (kind == 0) {                                                             // lambertian.rs:32-45
        f3 target = (p + n) + unit;
        att = (tex >= 0) ? sphere_uv_color(sc, tex, n) : mk(ma.x, ma.y, ma.z);
        nd = target - p;
    } else if (kind == 1) {                                                      // metal.rs:22-32
        f3 fuzz = unit * ma.w;
        att = mk(ma.x, ma.y, ma.z);
        nd = reflected + fuzz;
    } else {                                                                     // dielectric.rs:20-50
        const float ref_idx = ma.w;
        const f3 dir = ray.d;
        const float dt = dot(dir, n);
        f3 outward; float ni_over_nt, cosine;
        if (dt > 0.0f) { cosine = ref_idx * dt; outward = neg(n); ni_over_nt = ref_idx; }
        else { cosine = -dt; outward = n; ni_over_nt = ma.y; }                   // 1.0 / ref_idx (dielectric.rs:41), divided at upload
        f3 refracted;
        att = fill(1.0f);
        nd = reflected;
        if ((!EXT || kind == 2) && refract(dir, outward, ni_over_nt, refracted)) {
            float prob = schlick(cosine, ma.x);
            float r = rng_f32(rng);
            nd = (r < prob) ? reflected : refracted;
        }
    }
    out.o = p; out.d = normalized(nd);                                           // ray.rs:19-24 (discarded when absorbed)
    return alive;
}
// Sphere::record (sphere.rs:37-42) + scatter for a hit sphere
PTD bool scatter(const Scene &sc, const Ray &ray, const Hit &h, uint64_t &rng, f3 &att, Ray &out)
{
    const float4 sp = sc.prims[2 * h.prim];
    const float4 ma = sc.prims[2 * h.prim + 1];
    const int2 mb = make_int2(__float_as_int(sp.w) & 3, (__float_as_int(sp.w) >> 2) - 1);   // (kind, texture or -1)
    const f3 p = ray.o + ray.d * h.t;                                            // ray.rs:26-28
    const f3 n = normalized(p - mk(sp.x, sp.y, sp.z));
    return scatter_at<false>(sc, ray, p, n, mb.x, mb.y, ma, rng, att, out);
}
// Extension: the record of a sphere or a rect, the light's emitted term, then scatter.
//   prims[2 i] of a rect = (bits(flags), -, -, bits(kind | 1 << 30)); flags as in its Leaf (rect_step)
PTD bool scatter_ext(const Scene &sc, const Ray &ray, const Hit &h, uint64_t &rng, f3 &att, Ray &out, f3 &emitted, bool &light,
                     const uint32_t xf_lds = XF_GLOBAL)
{
    const float4 sp = sc.prims[2 * h.prim];
    const float4 ma = sc.prims[2 * h.prim + 1];
    const int tag = __float_as_int(sp.w);
    const int kind = tag & 3;
    f3 p, n;
    if (tag & (1 << 30)) {
        const int flags = __float_as_int(sp.x);
        const int axis = (flags >> 1) & 3, xf = (flags >> 4) - 1;
        f3 o = ray.o, d = ray.d;
        rect_object_ray(sc, flags, o, d, xf_lds);
        p = o + d * h.t;
        const float one = (flags & 8) ? -1.0f : 1.0f;
        n = mk(axis == 0 ? one : 0.0f, axis == 1 ? one : 0.0f, axis == 2 ? one : 0.0f);
        if (xf >= 0) {
            const Xform x = load_xform(sc, xf, xf_lds);
            p = rot_to_world(x, p) + x.off;
            n = rot_to_world(x, n);
        }
    } else {
        p = ray.o + ray.d * h.t;
        n = normalized(p - mk(sp.x, sp.y, sp.z));
    }
    light = kind == 3;
    emitted = mk(ma.x, ma.y, ma.z);                                              // read only when `light`
    return scatter_at<true>(sc, ray, p, n, kind, ((tag & 0x3fffffff) >> 2) - 1, ma, rng, att, out);
}
PTD f3 sky(const Ray &ray, f3 att)                                               // raytracer.rs:67-70
{
    float t = 0.5f * (ray.d.y + 1.0f);
    f3 color = (mk(1.0f, 1.0f, 1.0f) * (1.0f - t)) + (mk(0.5f, 0.7f, 1.0f) * t);
    return color * att;
}

#undef PTD
}  // inline namespace exactmath / fastmath
}  // namespace ptd
