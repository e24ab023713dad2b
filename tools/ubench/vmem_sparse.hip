// What does a 16-byte-per-lane gather (one node half per lane, as in the SPLIT run of a tree larger than the LDS
// prefix) cost the CU when only SOME lanes are enabled?  Four independent dependent-load chains per lane, 8 waves
// per SIMD, a table that sits in the L1 (16 KiB) or in the L2 (336 KiB); CU cycles per wave-instruction at 2.3 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
static float bits(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
__global__ void k(const float4 *__restrict__ tab, unsigned nmask, float *out, int iters, unsigned long long mask)
{
    const int lane = threadIdx.x & 63;
    unsigned i0 = (threadIdx.x * 7919u + blockIdx.x * 104729u) & nmask, i1 = (i0 * 31u + 7u) & nmask, i2 = (i0 * 17u + 3u) & nmask, i3 = (i0 * 13u + 1u) & nmask;
    float acc = 0.0f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {
            const float4 a = tab[i0], b = tab[i1], c = tab[i2], d = tab[i3];
            i0 = __float_as_uint(a.w) & nmask; i1 = __float_as_uint(b.w) & nmask; i2 = __float_as_uint(c.w) & nmask; i3 = __float_as_uint(d.w) & nmask;
            acc += a.x + b.y + c.z + d.x;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main()
{
    static const struct { unsigned long long m; const char *n; } MASKS[] = {
        {~0ull, "64"}, {0xffffffffull, "32"}, {0xffffull, "16"}, {0xffull, "8"}, {0xfull, "4"}, {1ull, "1"},
        {0x0101010101010101ull, "8 spread"}, {0x1111111111111111ull, "16 spread"}, {0x5555555555555555ull, "32 spread"}};
    const int blocks = 256 * 8, iters = 2000;
    float *out; hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (unsigned n : {1024u, 32768u}) {
        std::vector<float4> h(n);
        unsigned s = 12345u;
        for (unsigned i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = make_float4(1.0f, 2.0f, 3.0f, bits((s >> 8) % n)); }
        float4 *tab; hipMalloc(&tab, n * sizeof(float4)); hipMemcpy(tab, h.data(), n * sizeof(float4), hipMemcpyHostToDevice);
        printf("table of %u entries (%u KiB): CU cycles per wave-instruction, by enabled lanes\n", n, n * 16 / 1024);
        for (auto &mk : MASKS) {
            k<<<blocks, 256>>>(tab, n - 1, out, 10, mk.m); hipDeviceSynchronize();
            hipEventRecord(e0); k<<<blocks, 256>>>(tab, n - 1, out, iters, mk.m); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("  %-10s %7.2f\n", mk.n, ms * 1e-3 * 2.3e9 / ((double)iters * 4 * 32));
        }
        hipFree(tab);
    }
    return 0;
}
