// second VALU-rate table: single-instruction kinds via inline asm (no compiler fusion)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define BODY(ASM) \
    for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); } }
template <int KIND>
__global__ __launch_bounds__(512) void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a[8]; uint32_t b = seed + threadIdx.x, c = seed * 3 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed * (i + 1) + threadIdx.x;
    if (KIND == 0) BODY("v_min_f32 %0, %0, %1")
    if (KIND == 1) BODY("v_max_u32 %0, %0, %1")
    if (KIND == 2) BODY("v_cndmask_b32 %0, %0, %1, vcc")
    if (KIND == 3) BODY("v_bfe_u32 %0, %0, 3, 5")
    if (KIND == 4) BODY("v_lshlrev_b32 %0, 3, %0")
    if (KIND == 5) BODY("v_bcnt_u32_b32 %0, %0, %1")
    if (KIND == 6) BODY("v_mov_b32 %0, %1")
    if (KIND == 7) BODY("v_cvt_i32_f32 %0, %0")
    if (KIND == 8) BODY("v_med3_f32 %0, %0, %1, %2")
    if (KIND == 9) BODY("v_med3_i32 %0, %0, %1, %2")
    if (KIND == 10) BODY("v_add3_u32 %0, %0, %1, %2")
    if (KIND == 11) BODY("v_and_or_b32 %0, %0, %1, %2")
    if (KIND == 12) BODY("v_pk_add_u16 %0, %0, %1")
    if (KIND == 13) BODY("v_mad_u32_u24 %0, %0, %1, %2")
    if (KIND == 14) BODY("v_cmp_lt_f32 vcc, %0, %1")
    if (KIND == 15) BODY("v_cmp_lt_u32 vcc, %0, %1")
    if (KIND == 16) BODY("v_add_f32 %0, %0, %1")
    if (KIND == 17) BODY("v_xor_b32 %0, %0, %1")
    if (KIND == 18) BODY("v_sub_u32 %0, %0, %1")
    if (KIND == 19) BODY("v_mul_lo_u32 %0, %0, %1")
    if (KIND == 20) BODY("v_sqrt_f32 %0, %0")
    if (KIND == 21) BODY("v_pk_min_u16 %0, %0, %1")
    if (KIND == 22) BODY("v_max_f32 %0, %0, %1")
    if (KIND == 23) BODY("v_pk_mul_f32 %0, %0, %1")
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND> void run(const char *name)
{
    uint32_t *d; hipMalloc(&d, 256 * 4 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    k<KIND><<<256 * 4, 512>>>(d, 12345u, 16);
    hipEventRecord(e0); k<KIND><<<256 * 4, 512>>>(d, 12345u, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = 8.0 * iters * 8;
    printf("%-18s %.2f ns/instr/SIMD\n", name, ms * 1e6 / instr);
    hipFree(d);
}
int main()
{
    run<16>("v_add_f32"); run<0>("v_min_f32"); run<22>("v_max_f32"); run<8>("v_med3_f32"); run<14>("v_cmp_lt_f32"); run<7>("v_cvt_i32_f32"); run<20>("v_sqrt_f32");
    run<1>("v_max_u32"); run<2>("v_cndmask_b32"); run<3>("v_bfe_u32"); run<4>("v_lshlrev_b32"); run<5>("v_bcnt_u32_b32");
    run<6>("v_mov_b32"); run<9>("v_med3_i32"); run<10>("v_add3_u32"); run<11>("v_and_or_b32"); run<12>("v_pk_add_u16"); run<21>("v_pk_min_u16");
    run<13>("v_mad_u32_u24"); run<15>("v_cmp_lt_u32"); run<17>("v_xor_b32"); run<18>("v_sub_u32"); run<19>("v_mul_lo_u32");
    return 0;
}
