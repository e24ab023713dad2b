// Exhaustive check (all 2^32 fp32 bit patterns) of the short forms of 1/x, sqrt(x) and
// 1/sqrt(x) used by pt_device.hpp against the compiler's correctly rounded IEEE expansions
// (what the library used before and what the oracle's C code computes).  Prints, per
// function, the number of mismatching inputs inside the guard range and the first few.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o unary_exhaustive unary_exhaustive.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include "../../path_tracer_amd/csrc/pt_device.hpp"

__device__ __forceinline__ bool same(float a, float b)
{
    if (a != a && b != b) return true;                       // any NaN == any NaN
    return __float_as_uint(a) == __float_as_uint(b);
}
template <int F> __global__ void check(unsigned long long *bad, uint32_t *first, uint32_t base)
{
    const uint32_t bits = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float x = __uint_as_float(bits);
    float ref, got;
    if (F == 0) { ref = 1.0f / x; got = ptd::p_rcp(x); }
    else if (F == 1) { ref = __builtin_sqrtf(x); got = ptd::p_sqrt(x); }
    else { ref = 1.0f / __builtin_sqrtf(x); got = ptd::rsq(x); }
    if (!same(ref, got)) {
        unsigned long long n = atomicAdd(bad, 1ull);
        if (n < 8) first[n] = bits;
    }
}
template <int F> void run(const char *name)
{
    unsigned long long *d_bad, h_bad = 0; uint32_t *d_first, h_first[8];
    hipMalloc(&d_bad, 8); hipMalloc(&d_first, 32); hipMemset(d_bad, 0, 8); hipMemset(d_first, 0, 32);
    for (uint64_t base = 0; base < (1ull << 32); base += (1ull << 28))
        check<F><<<(1u << 28) / 256, 256>>>(d_bad, d_first, (uint32_t)base);
    hipDeviceSynchronize();
    hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(h_first, d_first, 32, hipMemcpyDeviceToHost);
    printf("%-8s mismatches over all 2^32 inputs: %llu", name, h_bad);
    for (unsigned i = 0; i < (h_bad < 8 ? h_bad : 8); i++) { float f; memcpy(&f, &h_first[i], 4); printf("  0x%08x(%g)", h_first[i], f); }
    printf("\n");
    hipFree(d_bad); hipFree(d_first);
}
int main()
{
    run<0>("1/x"); run<1>("sqrt"); run<2>("rsq");
    return 0;
}
