// Issue rate of scalar ALU instructions on gfx950, alone and mixed 1:2 with VALU, 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ void k(float *out, int iters, int seed)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float m = 1.0000001f;
    int s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {        // 8 SALU
            REP16(asm volatile("s_add_u32 %0, %0, 3\n s_xor_b32 %1, %1, %0\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, %2\n"
                               "s_add_u32 %0, %0, 7\n s_xor_b32 %1, %1, %0\n s_add_u32 %2, %2, 9\n s_xor_b32 %3, %3, %2\n"
                               : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");)
        } else if (KIND == 1) { // 8 VALU
            REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                               "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));)
        } else if (KIND == 2) { // 8 VALU + 4 SALU interleaved
            REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n s_add_u32 %4, %4, 3\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n s_xor_b32 %5, %5, %4\n"
                               "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n s_add_u32 %6, %6, 5\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n s_xor_b32 %7, %7, %6\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(m) : "scc");)
        } else {                // 8 VALU + 8 SALU interleaved
            REP16(asm volatile("v_mul_f32 %0, %0, %8\n s_add_u32 %4, %4, 3\n v_mul_f32 %1, %1, %8\n s_xor_b32 %5, %5, %4\n v_mul_f32 %2, %2, %8\n s_add_u32 %6, %6, 5\n v_mul_f32 %3, %3, %8\n s_xor_b32 %7, %7, %6\n"
                               "v_mul_f32 %0, %0, %8\n s_add_u32 %4, %4, 7\n v_mul_f32 %1, %1, %8\n s_xor_b32 %5, %5, %4\n v_mul_f32 %2, %2, %8\n s_add_u32 %6, %6, 9\n v_mul_f32 %3, %3, %8\n s_xor_b32 %7, %7, %6\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(m) : "scc");)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(s0 ^ s1 ^ s2 ^ s3);
}
template <int KIND> void run(const char *name, int w, double per_iter)
{
    float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    const int iters = 1000, blocks = 256 * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 10, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<blocks, 256>>>(d, iters, 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s waves/SIMD=%d  %.3f ms  %.2f cycles per group of (%s) per wave-slot-normalised SIMD at 2.2 GHz\n", name, w, ms,
           ms * 1e-3 * 2.2e9 / ((double)iters * 16 * w), name);
    (void)per_iter; hipFree(d);
}
int main()
{
    for (int w : {1, 4, 8}) { run<0>("8 salu", w, 8); run<1>("8 valu", w, 8); run<2>("8 valu + 4 salu", w, 12); run<3>("8 valu + 8 salu", w, 16); }
    return 0;
}
