// third VALU-rate table: operand-kind sensitivity (VGPR / SGPR / literal), logic ops, cndmask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define BODY(ASM) \
    for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "s"(sg)); } }
template <int KIND>
__global__ __launch_bounds__(512) void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a[8]; uint32_t b = seed + threadIdx.x, c = seed * 3 + threadIdx.x; uint32_t sg = seed * 7;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed * (i + 1) + threadIdx.x;
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(b), "v"(c) : "vcc");
    if (KIND == 0) BODY("v_and_b32 %0, %0, %1")
    if (KIND == 1) BODY("v_and_b32 %0, %3, %0")
    if (KIND == 2) BODY("v_and_b32 %0, 0x55555555, %0")
    if (KIND == 3) BODY("v_or_b32 %0, %0, %1")
    if (KIND == 4) BODY("v_add_u32 %0, %0, %1")
    if (KIND == 5) BODY("v_add_u32 %0, %3, %0")
    if (KIND == 6) BODY("v_lshrrev_b32 %0, 1, %0")
    if (KIND == 7) BODY("v_lshl_add_u32 %0, %0, 2, %1")
    if (KIND == 8) BODY("v_alignbit_b32 %0, %0, %1, 7")
    if (KIND == 9) BODY("v_bfi_b32 %0, %0, %1, %2")
    if (KIND == 10) BODY("v_cndmask_b32 %0, %0, %1, vcc")
    if (KIND == 11) BODY("v_add_f32 %0, %3, %0")
    if (KIND == 12) BODY("v_fma_f32 %0, %0, %3, %1")
    if (KIND == 13) BODY("v_mul_f32 %0, 0x3f8003f8, %0")
    if (KIND == 14) BODY("v_min_u32 %0, %0, %1")
    if (KIND == 15) BODY("v_sub_f32 %0, %0, %1")
    if (KIND == 16) BODY("v_xor_b32 %0, %3, %0")
    if (KIND == 17) BODY("v_readlane_b32 s20, %0, 3")
    if (KIND == 18) BODY("v_rcp_f32 %0, %0")
    if (KIND == 19) BODY("v_lshrrev_b64 %0, 3, %0")
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
    printf("%-26s %.2f ns/instr/SIMD\n", name, ms * 1e6 / (8.0 * iters * 8));
    hipFree(d);
}
int main()
{
    run<0>("v_and_b32 v,v"); run<1>("v_and_b32 s,v"); run<2>("v_and_b32 literal,v"); run<3>("v_or_b32 v,v"); run<16>("v_xor_b32 s,v");
    run<4>("v_add_u32 v,v"); run<5>("v_add_u32 s,v"); run<6>("v_lshrrev_b32 1,v"); run<7>("v_lshl_add_u32"); run<8>("v_alignbit_b32");
    run<9>("v_bfi_b32"); run<10>("v_cndmask_b32 vcc"); run<14>("v_min_u32 v,v");
    run<11>("v_add_f32 s,v"); run<15>("v_sub_f32 v,v"); run<12>("v_fma_f32 v,s,v"); run<13>("v_mul_f32 literal,v"); run<18>("v_rcp_f32");
    run<17>("v_readlane_b32"); 
    return 0;
}
