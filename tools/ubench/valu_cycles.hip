// Cycles (s_memtime, shader clock) per wave64 VALU instruction per SIMD on gfx950, with all
// 64 lanes, the low 32 lanes, the low 16 lanes, or scattered lanes enabled: does the SIMD skip
// passes whose lanes are all disabled?
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ void k(float *out, unsigned long long *cyc, int iters, unsigned long long mask)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0000001f;
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = 0, t1 = 0;
    if ((mask >> lane) & 1ull) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; i++) {
            if (KIND == 0) {
                REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
            } else {
                REP16(asm volatile("v_med3_f32 %0, %0, %8, %1\n v_med3_f32 %1, %1, %8, %2\n v_med3_f32 %2, %2, %8, %3\n v_med3_f32 %3, %3, %8, %4\n"
                             "v_med3_f32 %4, %4, %8, %5\n v_med3_f32 %5, %5, %8, %6\n v_med3_f32 %6, %6, %8, %7\n v_med3_f32 %7, %7, %8, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (lane == 0 && blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run(const char *name, int waves_per_simd, unsigned long long mask, const char *mname)
{
    float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    unsigned long long *c; hipMalloc(&c, 8);
    const int iters = 2000, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, c, 10, mask); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<blocks, 256>>>(d, c, iters, mask); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    double insts_per_simd = (double)iters * 128 * waves_per_simd;
    printf("%-10s %-10s waves/SIMD=%d  %.3f ms  %.2f memtime-cycles per wave-instr per SIMD (clock %.2f GHz)\n", name, mname,
           waves_per_simd, ms, (double)h / insts_per_simd, (double)h / (ms * 1e6));
    hipFree(d); hipFree(c);
}
int main()
{
    struct { unsigned long long m; const char *n; } masks[] = {
        {~0ull, "all64"}, {0xffffffffull, "low32"}, {0xffffull, "low16"}, {0xffff0000ull, "16..31"}, {0xffull, "low8"},
        {1ull, "lane0"}, {0x0000ffff0000ffffull, "2x16"}, {0x00ff00ff00ff00ffull, "4x8"}, {0x1111111111111111ull, "1per4"},
        {0x0001000100010001ull, "1per16"}, {0x8000400020001ull, "4 scattered"}, {0xfffffffffffffffeull, "all but 0"},
        {0x00000000ffff00ffull, "24 low"}, {0x0f0f0f0f0f0f0f0full, "nibbles"}};
    for (int w : {8}) {
        for (auto &mk : masks) run<0>("v_mul_f32", w, mk.m, mk.n);
        for (auto &mk : masks) run<1>("v_med3_f32", w, mk.m, mk.n);
    }
    return 0;
}
