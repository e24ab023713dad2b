// Measures the issue rate of plain / packed fp32 VALU ops on gfx950 (cycles per wave64 instruction per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ void k(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0000001f;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {
            REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (KIND == 1) {
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&m));)
        } else {
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND> void run(const char *name, int waves_per_simd)
{
    float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    const int iters = 2000, blocks = 256 * waves_per_simd;   // 256-thread blocks: one wave per SIMD each
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts_per_simd = (double)iters * 128 * waves_per_simd;   // wave-instructions issued on each SIMD
    printf("%-12s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / insts_per_simd);
    hipFree(d);
}
int main()
{
    for (int w : {1, 2, 4, 8}) { run<0>("v_mul_f32", w); run<2>("v_fma_f32", w); run<1>("v_pk_mul_f32", w); }
    return 0;
}
