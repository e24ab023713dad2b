// Issue cost table (cycles per wave64 instruction per SIMD, 4 waves/SIMD) for the VALU ops the path kernel uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define BODY(I0,I1,I2,I3) asm volatile(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I0 "\n" I1 "\n" I2 "\n" I3 \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1) : "v"(m), "v"(n), "s"(sm), "v"(dm) : "vcc");
template <int K>
__global__ void k(float *out, int iters, unsigned long long sm)
{
    float a0 = threadIdx.x * 0.001f + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, m = 1.0000001f, n = 0.5f;
    double d0 = a0, d1 = a1, dm = 1.0000001;
    for (int i = 0; i < iters; i++) {
        REP8(
        if (K == 0) { BODY("v_mul_f32 %0, %0, %6", "v_mul_f32 %1, %1, %6", "v_mul_f32 %2, %2, %6", "v_mul_f32 %3, %3, %6") }
        if (K == 1) { BODY("v_add_f32 %0, %0, %6", "v_add_f32 %1, %1, %6", "v_add_f32 %2, %2, %6", "v_add_f32 %3, %3, %6") }
        if (K == 2) { BODY("v_fma_f32 %0, %0, %6, %7", "v_fma_f32 %1, %1, %6, %7", "v_fma_f32 %2, %2, %6, %7", "v_fma_f32 %3, %3, %6, %7") }
        if (K == 3) { BODY("v_min_f32 %0, %0, %6", "v_max_f32 %1, %1, %6", "v_min_f32 %2, %2, %6", "v_max_f32 %3, %3, %6") }
        if (K == 4) { BODY("v_max3_f32 %0, %0, %6, %7", "v_min3_f32 %1, %1, %6, %7", "v_max3_f32 %2, %2, %6, %7", "v_min3_f32 %3, %3, %6, %7") }
        if (K == 5) { BODY("v_cmp_lt_f32 vcc, %0, %6", "v_cmp_lt_f32 vcc, %1, %6", "v_cmp_lt_f32 vcc, %2, %6", "v_cmp_lt_f32 vcc, %3, %6") }
        if (K == 6) { BODY("v_cndmask_b32 %0, %0, %6, vcc", "v_cndmask_b32 %1, %1, %6, vcc", "v_cndmask_b32 %2, %2, %6, vcc", "v_cndmask_b32 %3, %3, %6, vcc") }
        if (K == 7) { BODY("v_cndmask_b32 %0, %0, %6, %8", "v_cndmask_b32 %1, %1, %6, %8", "v_cndmask_b32 %2, %2, %6, %8", "v_cndmask_b32 %3, %3, %6, %8") }
        if (K == 8) { BODY("v_xor_b32 %0, %0, %6", "v_xor_b32 %1, %1, %6", "v_xor_b32 %2, %2, %6", "v_xor_b32 %3, %3, %6") }
        if (K == 9) { BODY("v_lshlrev_b64 %4, 13, %4", "v_lshlrev_b64 %5, 13, %5", "v_lshlrev_b64 %4, 7, %4", "v_lshlrev_b64 %5, 7, %5") }
        if (K == 10) { BODY("v_mul_lo_u32 %0, %0, %6", "v_mul_lo_u32 %1, %1, %6", "v_mul_lo_u32 %2, %2, %6", "v_mul_lo_u32 %3, %3, %6") }
        if (K == 11) { BODY("v_mul_f64 %4, %4, %9", "v_mul_f64 %5, %5, %9", "v_mul_f64 %4, %4, %9", "v_mul_f64 %5, %5, %9") }
        if (K == 12) { BODY("v_rcp_f32 %0, %0", "v_rcp_f32 %1, %1", "v_rcp_f32 %2, %2", "v_rcp_f32 %3, %3") }
        if (K == 13) { BODY("v_sqrt_f32 %0, %0", "v_sqrt_f32 %1, %1", "v_sqrt_f32 %2, %2", "v_sqrt_f32 %3, %3") }
        if (K == 14) { BODY("v_div_scale_f32 %0, vcc, %0, %6, %0", "v_div_scale_f32 %1, vcc, %1, %6, %1", "v_div_scale_f32 %2, vcc, %2, %6, %2", "v_div_scale_f32 %3, vcc, %3, %6, %3") }
        if (K == 15) { BODY("v_div_fmas_f32 %0, %0, %6, %7", "v_div_fmas_f32 %1, %1, %6, %7", "v_div_fmas_f32 %2, %2, %6, %7", "v_div_fmas_f32 %3, %3, %6, %7") }
        if (K == 16) { BODY("v_div_fixup_f32 %0, %0, %6, %7", "v_div_fixup_f32 %1, %1, %6, %7", "v_div_fixup_f32 %2, %2, %6, %7", "v_div_fixup_f32 %3, %3, %6, %7") }
        if (K == 17) { BODY("v_mov_b32 %0, %6", "v_mov_b32 %1, %6", "v_mov_b32 %2, %6", "v_mov_b32 %3, %6") }
        if (K == 18) { BODY("v_fmac_f32 %0, %6, %7", "v_fmac_f32 %1, %6, %7", "v_fmac_f32 %2, %6, %7", "v_fmac_f32 %3, %6, %7") }
        if (K == 19) { BODY("v_pk_mul_f32 %4, %4, %9", "v_pk_mul_f32 %5, %5, %9", "v_pk_add_f32 %4, %4, %9", "v_pk_add_f32 %5, %5, %9") }
        if (K == 21) { BODY("v_add_u32 %0, %0, %6", "v_add_u32 %1, %1, %6", "v_add_u32 %2, %2, %6", "v_add_u32 %3, %3, %6") }
        if (K == 22) { BODY("v_lshl_add_u64 %4, %4, 4, %5", "v_lshl_add_u64 %5, %5, 4, %4", "v_lshl_add_u64 %4, %4, 4, %5", "v_lshl_add_u64 %5, %5, 4, %4") }
        if (K == 24) { BODY("v_min_i32 %0, %0, %6", "v_max_i32 %1, %1, %6", "v_min_u32 %2, %2, %6", "v_max_u32 %3, %3, %6") }
        if (K == 25) { BODY("v_cmp_lt_i32 vcc, %0, %6", "v_cmp_lt_u32 vcc, %1, %6", "v_cmp_lt_i32 vcc, %2, %6", "v_cmp_lt_u32 vcc, %3, %6") }
        if (K == 26) { BODY("v_and_b32 %0, %0, %6", "v_or_b32 %1, %1, %6", "v_lshlrev_b32 %2, 1, %2", "v_lshrrev_b32 %3, 1, %3") }
        if (K == 27) { BODY("v_med3_f32 %0, %0, %6, %7", "v_med3_f32 %1, %1, %6, %7", "v_med3_f32 %2, %2, %6, %7", "v_med3_f32 %3, %3, %6, %7") }
        if (K == 28) { BODY("v_sub_f32 %0, %0, %6", "v_sub_f32 %1, %6, %1", "v_sub_f32 %2, %2, %6", "v_sub_f32 %3, %6, %3") }
        if (K == 29) { BODY("v_mul_f32 %0, %0, %6\n v_mul_f32 %1, %1, %6", "v_min_f32 %2, %2, %6", "v_mul_f32 %0, %0, %6\n v_mul_f32 %1, %1, %6", "v_max_f32 %3, %3, %6") }
        if (K == 30) { BODY("v_bfe_u32 %0, %0, 3, 5", "v_bfi_b32 %1, %1, %6, %7", "v_alignbit_b32 %2, %2, %6, 7", "v_perm_b32 %3, %3, %6, %7") }
        if (K == 31) { BODY("v_add_f64 %4, %4, %9", "v_add_f64 %5, %5, %9", "v_fma_f64 %4, %4, %9, %9", "v_fma_f64 %5, %5, %9, %9") }
        if (K == 23) { BODY("v_cvt_f32_u32 %0, %0", "v_cvt_f32_u32 %1, %1", "v_cvt_f32_u32 %2, %2", "v_cvt_f32_u32 %3, %3") }
        )
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1);
}
__global__ void clk(unsigned long long *o) { unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); float a = threadIdx.x; for (int i = 0; i < 200000; i++) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a)); unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); if (threadIdx.x == 0 && blockIdx.x == 0) { o[0] = t1 - t0; o[1] = r1 - r0; } if (a == 12345.f) o[2] = 1; }
static float *d;
template <int K> void run(const char *name)
{
    const int iters = 1000, wps = 4, blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<K><<<blocks, 256>>>(d, 10, 0x5555555555555555ull); hipDeviceSynchronize();
    hipEventRecord(e0); k<K><<<blocks, 256>>>(d, iters, 0x5555555555555555ull); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * wps));
}
int main()
{
    hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    { unsigned long long *o, h[2]; hipMalloc(&o, 64); clk<<<1024, 256>>>(o); hipMemcpy(h, o, 16, hipMemcpyDeviceToHost); printf("shader clock under VALU load: %.0f MHz (memtime/memrealtime x 100 MHz)\n", (double)h[0] / (double)h[1] * 100.0); }
    run<0>("v_mul_f32"); run<1>("v_add_f32"); run<2>("v_fma_f32"); run<18>("v_fmac_f32"); run<3>("v_min/max_f32"); run<4>("v_min3/max3_f32");
    run<5>("v_cmp_lt_f32 vcc"); run<6>("v_cndmask_b32 vcc"); run<7>("v_cndmask_b32 sgpr"); run<8>("v_xor_b32"); run<21>("v_add_u32");
    run<9>("v_lshlrev_b64"); run<22>("v_lshl_add_u64"); run<10>("v_mul_lo_u32"); run<11>("v_mul_f64"); run<12>("v_rcp_f32"); run<13>("v_sqrt_f32");
    run<14>("v_div_scale_f32"); run<15>("v_div_fmas_f32"); run<16>("v_div_fixup_f32"); run<17>("v_mov_b32"); run<19>("v_pk_mul/add_f32");
    run<23>("v_cvt_f32_u32"); run<24>("v_min/max_i32/u32"); run<25>("v_cmp_lt_i32/u32"); run<26>("v_and/or/shl/shr_b32"); run<27>("v_med3_f32"); run<28>("v_sub_f32"); run<29>("4 mul + 2 minmax (6 instr)"); run<30>("bfe/bfi/alignbit/perm"); run<31>("v_add_f64/v_fma_f64");
    return 0;
}
