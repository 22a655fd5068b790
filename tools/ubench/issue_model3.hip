// issue_model2.hip — second pass of the gfx950 issue model: per-instruction cost classes.
// 32 independent instructions per loop trip (loop overhead < 10 %), >= 0.3 ms per launch, wall time
// only, at k = 1, 2, 3, 4, 8 waves per SIMD.  Output: ns per wave-instruction per SIMD.  A normal
// VALU op reads ~2.0-2.2 ns at k >= 3 (4 cycles of a ~2 GHz clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define BODY32(ASM)                                                                     \
    for (int it = 0; it < iters; ++it) {                                                \
        _Pragma("unroll") for (int i = 0; i < 32; ++i) {                                \
            asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "s"(sg));                   \
        }                                                                               \
    }
#define BODY32Q(ASM)                                                                    \
    for (int it = 0; it < iters; ++it) {                                                \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                \
            asm volatile(ASM : "+v"(q[i]) : "v"(b), "v"(c), "s"(sg));                   \
        }                                                                               \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                \
            asm volatile(ASM : "+v"(q[i]) : "v"(b), "v"(c), "s"(sg));                   \
        }                                                                               \
    }

extern __shared__ uint32_t lds[];

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a[32]; uint32_t b = seed + threadIdx.x, c = seed * 3 + threadIdx.x; uint32_t sg = seed * 7;
    unsigned long long q[16];
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = seed * (i + 1) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) q[i] = 0x0123456789abcdefull * (a[i] | 1u);
    uint32_t la = (threadIdx.x * 8u) & 0x3ffu;
    lds[threadIdx.x] = seed; lds[threadIdx.x + 256] = seed;
    __syncthreads();
    asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cmp_lt_u32 s[20:21], %1, %0" :: "v"(b), "v"(c) : "vcc", "s20", "s21");
    if (KIND == 0) BODY32("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc")
    if (KIND == 1) BODY32("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc")
    if (KIND == 2) BODY32("v_cmp_lt_f32 vcc, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
    if (KIND == 3) BODY32("v_cmp_lt_f32 vcc, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
    if (KIND == 4) BODY32("v_cndmask_b32_e64 %0, %0, %1, vcc")
    if (KIND == 5) BODY32("v_cmp_lt_f32 s[22:23], %0, %1\n\tv_cndmask_b32 %0, %0, %2, s[22:23]\n\tv_cndmask_b32 %0, %0, %1, s[22:23]")
    if (KIND == 6) BODY32("v_add_f32 %0, %0, %1\n\tv_fma_f32 %0, %0, %1, %2")
    if (KIND == 7) BODY32("v_add_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_perm_b32 %0, %0, %1, %2")
    if (KIND == 8) BODY32("v_add_u32 %0, %0, %1")
    if (KIND == 9) BODY32("v_fma_f32 %0, %0, %1, %2")
    if (KIND == 10) BODY32("v_lshlrev_b32 %0, %1, %0")
    if (KIND == 11) BODY32("v_lshlrev_b32 %0, 3, %0")
    if (KIND == 12) BODY32("v_lshrrev_b32 %0, %1, %0")
    if (KIND == 13) BODY32("v_ashrrev_i32 %0, 3, %0")
    if (KIND == 14) BODY32("v_sub_f32 %0, %0, %1")
    if (KIND == 15) BODY32("v_mul_f32 %0, 0x3f8003f8, %0")
    if (KIND == 16) BODY32("v_add_f32 %0, %3, %0")
    if (KIND == 17) BODY32("v_max_u32 %0, %0, %1")
    if (KIND == 18) BODY32("v_subrev_u32 %0, %0, %1")
    if (KIND == 19) BODY32("v_not_b32 %0, %0")
    if (KIND == 20) BODY32("v_bfe_u32 %0, %0, 3, 7")
    if (KIND == 21) BODY32("v_add_co_u32 %0, vcc, %0, %1")
    if (KIND == 22) BODY32("v_xor_b32 %0, 0x80000000, %0")
    if (KIND == 23) BODY32("v_or_b32 %0, 3, %0")
    if (KIND == 24) BODY32("v_add_u32 %0, 3, %0")
    if (KIND == 25) BODY32("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2")
    if (KIND == 26) BODY32("v_and_b32 %0, 0x0f0f0f0f, %0\n\tv_bcnt_u32_b32 %0, %0, %1")
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) r ^= a[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) r ^= (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r ^ sg;
}

template <int KIND> void run(const char *name, double per_trip)
{
    const int ks[7] = {1, 2, 3, 4, 5, 6, 8};
    printf("%-52s", name);
    for (int ki = 0; ki < 7; ++ki) {
        const int kw = ks[ki];
        const int blocks = 256 * kw;
        const size_t lds_bytes = (size_t)(160 * 1024 / kw - 1024) & ~(size_t)1023;
        uint32_t *d;
        (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        const int iters = 2048 / kw;
        k<KIND><<<blocks, 256, lds_bytes>>>(d, 12345u, 8);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0); k<KIND><<<blocks, 256, lds_bytes>>>(d, 12345u, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr_simd = (double)iters * per_trip * kw;
        printf(" k%d %5.2f", kw, (ms * 1e6 - 6000.0) / instr_simd);
        (void)hipFree(d);
    }
    printf("\n");
    fflush(stdout);
}
int main()
{
    printf("ns per group (one asm block, as named) per SIMD, wall time\n");
    run<0>("v_cmp vcc; 2x v_cndmask vcc (per group of 3)", 32);
    run<1>("v_cmp vcc; 4x v_cndmask vcc (per group of 5)", 32);
    run<2>("v_cmp vcc; v_add_u32; v_cndmask vcc (per group of 3)", 32);
    run<3>("v_cmp vcc; 4x v_add_u32; v_cndmask vcc (group of 6)", 32);
    run<4>("v_cndmask_e64 with vcc operand", 32);
    run<5>("v_cmp s[22:23]; 2x v_cndmask s[22:23] (group of 3)", 32);
    run<6>("v_add_f32; v_fma_f32 (per pair)", 32);
    run<7>("v_add_f32; v_mul_f32; v_fma_f32; v_perm (group of 4)", 32);
    run<8>("v_add_u32 v,v", 32);
    run<9>("v_fma_f32", 32);
    run<10>("v_lshlrev_b32 v,v", 32);
    run<11>("v_lshlrev_b32 imm", 32);
    run<12>("v_lshrrev_b32 v,v", 32);
    run<13>("v_ashrrev_i32 imm", 32);
    run<14>("v_sub_f32", 32);
    run<15>("v_mul_f32 literal", 32);
    run<16>("v_add_f32 sgpr", 32);
    run<17>("v_max_u32", 32);
    run<18>("v_subrev_u32", 32);
    run<19>("v_not_b32", 32);
    run<20>("v_bfe_u32", 32);
    run<21>("v_add_co_u32 vcc", 32);
    run<22>("v_xor_b32 literal", 32);
    run<23>("v_or_b32 inline const", 32);
    run<24>("v_add_u32 inline const", 32);
    run<25>("v_mul_f32 v,v then v_add_f32 same reg (dep pair)", 32);
    run<26>("v_and_b32; v_bcnt (pair)", 32);
    return 0;
}
