// Exploration: where do the short forms differ from the IEEE expansions?  (all 2^32 inputs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float rcp_fast(float s)
{
    float r = __builtin_amdgcn_rcpf(s);
    float e = __builtin_fmaf(-s, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float sqrt_fast(float x)
{
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_fast2(float x)      // one more coupled iteration
{
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h);
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ bool same(float a, float b) { return (a != a && b != b) || __float_as_uint(a) == __float_as_uint(b); }
template <int F> __global__ void check(unsigned long long *bad, uint32_t *lohi, uint32_t base)
{
    const uint32_t bits = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float x = __uint_as_float(bits);
    float ref, got;
    if (F == 0) { ref = 1.0f / x; got = rcp_fast(x); }
    else if (F == 1) { ref = __builtin_sqrtf(x); got = sqrt_fast(x); }
    else if (F == 2) { ref = __builtin_sqrtf(x); got = sqrt_fast2(x); }
    else if (F == 3) { ref = 1.0f / __builtin_sqrtf(x); got = rcp_fast(sqrt_fast(x)); }
    else { ref = 1.0f / __builtin_sqrtf(x); got = rcp_fast(sqrt_fast2(x)); }
    const uint32_t mag = bits & 0x7fffffffu;
    const bool mid = mag >= 0x10000000u && mag <= 0x6f000000u && (F == 0 || bits < 0x80000000u);   // 2^-95 .. 2^95
    if (!same(ref, got)) {
        atomicAdd(&bad[mid ? 0 : 1], 1ull);
        if (mid) { atomicMin(&lohi[0], mag); atomicMax(&lohi[1], mag); unsigned long long n = atomicAdd(&bad[2], 1ull); if (n < 6) lohi[2 + n] = bits; }
        else if (mag > 0x00800000u && mag < 0x7f800000u && (F == 0 || bits < 0x80000000u)) { atomicMin(&lohi[8], mag); atomicMax(&lohi[9], mag); }
    }
}
template <int F> void run(const char *name)
{
    unsigned long long *d_bad, h_bad[3]; uint32_t *d_lohi, h[10];
    hipMalloc(&d_bad, 24); hipMalloc(&d_lohi, 40); hipMemset(d_bad, 0, 24);
    uint32_t init[10] = {0xffffffffu, 0, 0, 0, 0, 0, 0, 0, 0xffffffffu, 0}; hipMemcpy(d_lohi, init, 40, hipMemcpyHostToDevice);
    for (uint64_t base = 0; base < (1ull << 32); base += (1ull << 28)) check<F><<<(1u << 28) / 256, 256>>>(d_bad, d_lohi, (uint32_t)base);
    hipDeviceSynchronize();
    hipMemcpy(h_bad, d_bad, 24, hipMemcpyDeviceToHost); hipMemcpy(h, d_lohi, 40, hipMemcpyDeviceToHost);
    printf("%-22s mismatches: %llu in |x| in [2^-95, 2^95], %llu outside; mid range lo 0x%08x hi 0x%08x; outside-normal lo 0x%08x hi 0x%08x; first:", name, h_bad[0], h_bad[1], h[0], h[1], h[8], h[9]);
    for (int i = 0; i < 6 && i < (int)h_bad[2]; i++) printf(" 0x%08x", h[2 + i]);
    printf("\n");
}
int main()
{
    run<0>("rcp: rcp+2fma"); run<1>("sqrt: rsq+mul,mul,2fma"); run<2>("sqrt: coupled"); run<3>("rsq: rcp(sqrt_fast)"); run<4>("rsq: rcp(sqrt coupled)");
    return 0;
}
