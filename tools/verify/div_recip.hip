// x / a from a precomputed ra = RN(1/a):  q = x*ra; two fused residual corrections.
// Not exhaustively checkable (2^64 operand pairs): random pairs + pairs built to land next to
// rounding midpoints, against the IEEE division.  Prints mismatch counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../path_tracer_amd/csrc/pt_device.hpp"
__device__ __forceinline__ uint64_t mix(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
__device__ __forceinline__ float div_r(float x, float a, float ra)
{
    float q = x * ra;
    float r = __builtin_fmaf(-a, q, x);
    q = __builtin_fmaf(r, ra, q);
    r = __builtin_fmaf(-a, q, x);
    return __builtin_fmaf(r, ra, q);
}
__device__ __forceinline__ float div_r1(float x, float a, float ra)   // one correction only: the control
{
    float q = x * ra;
    float r = __builtin_fmaf(-a, q, x);
    return __builtin_fmaf(r, ra, q);
}
__global__ void check(unsigned long long *bad, uint32_t *first, uint64_t seed, int mode, int weak)
{
    const uint64_t id = seed + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < 64; k++) {
        uint64_t h = mix(id * 64 + k);
        // a: mantissa random (biased to extremes), exponent in [-32, 32] (regular_ray)
        uint32_t am = (uint32_t)h & 0x7fffffu;
        if ((h >> 60) == 0) am = 0x7fffffu - (am & 0xff);
        if ((h >> 60) == 1) am = am & 0xff;
        int ae = (int)((h >> 23) % 65) - 32;
        float a = __uint_as_float(((uint32_t)(ae + 127) << 23) | am);
        uint64_t g = mix(h);
        float x;
        if (mode == 0) {
            uint32_t xm = (uint32_t)g & 0x7fffffu; int xe = (int)((g >> 23) % 121) - 60;
            x = __uint_as_float(((uint32_t)(g >> 63) << 31) | ((uint32_t)(xe + 127) << 23) | xm);
        } else {
            // x = RN(a * (m + half an ulp)) +- few ulps: quotient next to a midpoint
            uint32_t qm = (uint32_t)g & 0x7fffffu; int qe = (int)((g >> 23) % 61) - 9;
            double m = (double)__uint_as_float(((uint32_t)(qe + 127) << 23) | qm);
            double mid = m * (1.0 + 0.0) + (double)__uint_as_float((uint32_t)(qe + 127 - 24) << 23);
            float x0 = (float)((double)a * mid);
            int d = (int)((g >> 40) % 5) - 2;
            x = __uint_as_float(__float_as_uint(x0) + d);
            if ((g >> 62) & 1) x = -x;
        }
        const float ra = ptd::p_rcp(a);
        const float ref = x / a, got = weak ? div_r1(x, a, ra) : div_r(x, a, ra);
        if (__float_as_uint(ref) != __float_as_uint(got)) {
            unsigned long long n = atomicAdd(bad, 1ull);
            if (n < 4) { first[2 * n] = __float_as_uint(x); first[2 * n + 1] = __float_as_uint(a); }
        }
    }
}
int main()
{
    unsigned long long *d_bad, h_bad; uint32_t *d_first, h_first[8];
    hipMalloc(&d_bad, 8); hipMalloc(&d_first, 32);
    for (int weak = 0; weak < 2; weak++)
    for (int mode = 0; mode < 2; mode++) {
        hipMemset(d_bad, 0, 8); hipMemset(d_first, 0, 32);
        for (int it = 0; it < 64; it++) check<<<1 << 16, 256>>>(d_bad, d_first, (uint64_t)it << 32, mode, weak);
        hipDeviceSynchronize();
        hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(h_first, d_first, 32, hipMemcpyDeviceToHost);
        printf("%s mode %d (%s): %llu mismatches in %.3g pairs;", weak ? "[control: one correction]" : "[two corrections]", mode, mode ? "near midpoints" : "random", h_bad, 64.0 * 65536 * 256 * 64);
        for (unsigned i = 0; i < (h_bad < 4 ? h_bad : 4); i++) printf(" x=0x%08x a=0x%08x", h_first[2 * i], h_first[2 * i + 1]);
        printf("\n");
    }
    return 0;
}
