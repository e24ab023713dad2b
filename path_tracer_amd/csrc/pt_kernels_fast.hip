// pt_kernels_fast.hip -- the FAST build of the path kernel (pt_set_mode(ctx, PT_FAST), include/pt_amd.h).
//
// The same source as pt_kernels.hip, compiled with floating-point contraction ON and PT_FAST_MODE
// defined: a * b + c becomes one fused multiply-add wherever the compiler finds it (dot products, ray
// points, reflect / refract), the slab test takes one v_fma_f32 per plane (15 instead of 21 vector
// instructions per node; a wave with a lane whose |o * inv| exceeds 2^12 keeps the exact test for that
// walk), 1/x, sqrt and 1/sqrt are the hardware estimates as they are, and Sphere::trace multiplies by
// 1 / dot(d, d) instead of dividing -- its discriminant stays un-fused (PT_FAST_LEAF in pt_device.hpp:
// fusing it moves every silhouette).  Camera rays are the EXACT ones.  None of this is the reference's
// rounding: FAST results are NOT bit-equal to the oracle's.  They meet BASELINE's north-star tolerance
// (>= 99.9 % of pixels within 1/255 after tonemapping at equal spp and seed, from 256 spp up -- 99.8 % at
// 100 spp, where one path that goes another way moves a pixel by more than 1/255; converged RMSE < 1e-3:
// tests/test_gpu_fast.py).  The default mode, every other kernel, the primary-hit AOV (bit-exact hit ids)
// and the extension's scenes always use the EXACT build.  Only ptk_fast::launch_path comes out of this file.
// Exact ties in t go to the candidate the walk meets first (no rank rule: PT_NO_TIE), whatever the walk order.
#pragma clang fp contract(fast)
#define PT_FAST_MODE 1
#define PT_NO_TIE 1
#include "pt_kernels.hip"
