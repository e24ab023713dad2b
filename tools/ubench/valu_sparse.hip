// Does the cost of a wave64 VALU instruction on gfx950 depend on HOW MANY lanes are enabled?
// (it does: see DESIGN.md "sparse exec").  Kernel time for 8 waves/SIMD, per instruction kind
// and exec mask, as cycles per wave-instruction per SIMD at the clock measured by s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY8(OP) asm volatile(OP(%0, %1) OP(%1, %2) OP(%2, %3) OP(%3, %4) OP(%4, %5) OP(%5, %6) OP(%6, %7) OP(%7, %0) \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(mi));
#define S(x) #x
#define OP_MUL(d, s)   "v_mul_f32 " S(d) ", " S(d) ", %8\n"
#define OP_MUL64(d, s) "v_mul_f32_e64 " S(d) ", " S(d) ", %8\n"
#define OP_FMA(d, s)   "v_fma_f32 " S(d) ", " S(d) ", %8, " S(s) "\n"
#define OP_MED3(d, s)  "v_med3_f32 " S(d) ", " S(d) ", %8, " S(s) "\n"
#define OP_MAX(d, s)   "v_max_f32 " S(d) ", " S(d) ", " S(s) "\n"
#define OP_RCP(d, s)   "v_rcp_f32 " S(d) ", " S(d) "\n"
#define OP_ADDU(d, s)  "v_add_u32 " S(d) ", " S(d) ", %9\n"
#define OP_XOR(d, s)   "v_xor_b32 " S(d) ", " S(d) ", " S(s) "\n"
#define OP_LSHL(d, s)  "v_lshlrev_b32 " S(d) ", 1, " S(d) "\n"
#define OP_CMP(d, s)   "v_cmp_lt_f32 vcc, " S(d) ", " S(s) "\n"
#define OP_CNDM(d, s)  "v_cndmask_b32 " S(d) ", " S(d) ", " S(s) ", vcc\n"
#define OP_SUB(d, s)   "v_sub_f32 " S(d) ", " S(d) ", " S(s) "\n"
#define OP_FMAC(d, s)  "v_fmac_f32 " S(d) ", %8, " S(s) "\n"
#define OP_DIVFIX(d, s) "v_div_fixup_f32 " S(d) ", " S(d) ", %8, " S(s) "\n"
#define OP_SQRT(d, s)  "v_sqrt_f32 " S(d) ", " S(d) "\n"
#define OP_MOV(d, s)   "v_mov_b32 " S(d) ", " S(s) "\n"
#define OP_MULLO(d, s) "v_mul_lo_u32 " S(d) ", " S(d) ", " S(s) "\n"
#define OP_FMA64(d, s) "v_fma_f64 %[d0], %[d0], %[d1], %[d1]\n"
template <int KIND>
__global__ void k(float *out, unsigned long long *cyc, int iters, unsigned long long mask)
{
    float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0000001f; int mi = 3;
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = 0, t1 = 0;
    if ((mask >> lane) & 1ull) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; i++) {
            if (KIND == 0) { REP16(BODY8(OP_MUL)) }
            if (KIND == 1) { REP16(BODY8(OP_MUL64)) }
            if (KIND == 2) { REP16(BODY8(OP_FMA)) }
            if (KIND == 3) { REP16(BODY8(OP_MED3)) }
            if (KIND == 4) { REP16(BODY8(OP_MAX)) }
            if (KIND == 5) { REP16(BODY8(OP_RCP)) }
            if (KIND == 6) { REP16(BODY8(OP_ADDU)) }
            if (KIND == 7) { REP16(BODY8(OP_XOR)) }
            if (KIND == 8) { REP16(BODY8(OP_LSHL)) }
            if (KIND == 9) { REP16(BODY8(OP_CMP)) }
            if (KIND == 10) { REP16(BODY8(OP_CNDM)) }
            if (KIND == 11) { REP16(BODY8(OP_SUB)) }
            if (KIND == 12) { REP16(BODY8(OP_FMAC)) }
            if (KIND == 13) { REP16(BODY8(OP_DIVFIX)) }
            if (KIND == 14) { REP16(BODY8(OP_SQRT)) }
            if (KIND == 15) { REP16(BODY8(OP_MOV)) }
            if (KIND == 16) { REP16(BODY8(OP_MULLO)) }
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (t1 != 0 && blockIdx.x == 0 && (threadIdx.x >> 6) == 0) cyc[0] = t1 - t0;
}
static const struct { unsigned long long m; const char *n; } MASKS[] = {
    {~0ull, "64"}, {0xffffffffull, "32"}, {0xffffffull, "24"}, {0x1ffffull, "17"}, {0xffffull, "16"}, {0xffull, "8"}, {1ull, "1"},
    {0x0101010101010101ull, "8 spread"}, {0x1111111111111111ull, "16 spread"}, {0x1111111133333333ull, "24 spread"}};
template <int KIND> void run(const char *name)
{
    float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    unsigned long long *c; hipMalloc(&c, 8);
    const int iters = 500, w = 8, blocks = 256 * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-16s", name);
    for (auto &mk : MASKS) {
        k<KIND><<<blocks, 256>>>(d, c, 10, mk.m); hipDeviceSynchronize();
        hipEventRecord(e0); k<KIND><<<blocks, 256>>>(d, c, iters, mk.m); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf(" %6.2f", ms * 1e-3 * 2.2e9 / ((double)iters * 128 * w));
    }
    printf("\n");
    hipFree(d); hipFree(c);
}
int main()
{
    printf("cycles per wave-instruction per SIMD at 2.2 GHz, 8 waves/SIMD; columns = enabled lanes:\n%-16s", "");
    for (auto &mk : MASKS) printf(" %6s", mk.n);
    printf("\n");
    run<0>("v_mul_f32"); run<1>("v_mul_f32_e64"); run<11>("v_sub_f32"); run<4>("v_max_f32"); run<2>("v_fma_f32"); run<12>("v_fmac_f32");
    run<3>("v_med3_f32"); run<13>("v_div_fixup_f32"); run<5>("v_rcp_f32"); run<14>("v_sqrt_f32"); run<6>("v_add_u32"); run<7>("v_xor_b32");
    run<8>("v_lshlrev_b32"); run<16>("v_mul_lo_u32"); run<9>("v_cmp_lt_f32"); run<10>("v_cndmask_b32"); run<15>("v_mov_b32");
    return 0;
}
