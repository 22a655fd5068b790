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
    if (KIND == 0) BODY32("v_add_u32 %0, %0, %1")
    if (KIND == 1) BODY32("v_add_u32 %0, %3, %0")
    if (KIND == 2) BODY32("v_and_b32 %0, %0, %1")
    if (KIND == 3) BODY32("v_xor_b32 %0, %0, %1")
    if (KIND == 4) BODY32("v_lshlrev_b32 %0, 3, %0")
    if (KIND == 5) BODY32("v_sub_u32 %0, %0, %1")
    if (KIND == 6) BODY32("v_min_u32 %0, %0, %1")
    if (KIND == 7) BODY32("v_mov_b32 %0, %1")
    if (KIND == 8) BODY32("v_add_f32 %0, %0, %1")
    if (KIND == 9) BODY32("v_mul_f32 %0, %0, %1")
    if (KIND == 10) BODY32("v_fma_f32 %0, %0, %1, %2")
    if (KIND == 11) BODY32("v_max_f32 %0, %0, %1")
    if (KIND == 12) BODY32("v_cmp_lt_f32 vcc, %0, %1")
    if (KIND == 13) BODY32("v_cmp_lt_f32 s[22:23], %0, %1")
    if (KIND == 14) BODY32("v_cndmask_b32 %0, %0, %1, vcc")
    if (KIND == 15) BODY32("v_cndmask_b32 %0, %0, %1, s[20:21]")
    if (KIND == 16) BODY32("v_cndmask_b32 %0, %1, %2, vcc")
    if (KIND == 17) BODY32("v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, vcc")
    if (KIND == 18) BODY32("v_cmp_lt_f32 s[22:23], %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, s[22:23]")
    if (KIND == 19) BODY32("v_addc_co_u32 %0, vcc, %0, %1, vcc")
    if (KIND == 20) BODY32("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
    if (KIND == 21) BODY32("v_pk_add_u16 %0, %0, %1")
    if (KIND == 22) BODY32("v_pk_lshlrev_b16 %0, 4, %0")
    if (KIND == 23) BODY32("v_lshl_add_u32 %0, %0, 3, %1")
    if (KIND == 24) BODY32("v_and_or_b32 %0, %0, %1, %2")
    if (KIND == 25) BODY32("v_bfi_b32 %0, %0, %1, %2")
    if (KIND == 26) BODY32("v_alignbit_b32 %0, %0, %1, 8")
    if (KIND == 27) BODY32("v_mbcnt_lo_u32_b32 %0, %0, %1")
    if (KIND == 28) BODY32("v_cvt_i32_f32 %0, %0")
    if (KIND == 29) BODY32("v_cvt_f32_ubyte0 %0, %0")
    if (KIND == 30) BODY32("v_floor_f32 %0, %0")
    if (KIND == 31) BODY32("v_sad_u8 %0, %0, %1, %2")
    if (KIND == 32) BODY32("v_dot4_u32_u8 %0, %0, %1, %2")
    if (KIND == 33) BODY32("v_max3_u32 %0, %0, %1, %2")
    if (KIND == 34) BODY32("v_sqrt_f32 %0, %0")
    if (KIND == 35) BODY32("v_rsq_f32 %0, %0")
    if (KIND == 36) BODY32Q("v_pk_fma_f32 %0, %0, %0, %0")
    if (KIND == 37) BODY32Q("v_pk_add_f32 %0, %0, %0")
    if (KIND == 38) BODY32Q("v_lshlrev_b64 %0, 3, %0")
    if (KIND == 39) BODY32("v_bcnt_u32_b32 %0, %0, %1")
    if (KIND == 40) BODY32("v_perm_b32 %0, %0, %1, %2")
    if (KIND == 41) BODY32("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
    if (KIND == 42) BODY32("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
    if (KIND == 43) BODY32("v_cmp_eq_u32 vcc, %0, %1")
    if (KIND == 44) BODY32("v_sub_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %2")          // 2 normal fp ops
    if (KIND == 45) BODY32("v_add_u32 %0, %0, %1\n\ts_add_u32 %3, %3, 7")             // cheap valu + salu
    if (KIND == 46) BODY32("v_fma_f32 %0, %0, %1, %2\n\ts_add_u32 %3, %3, 7")          // normal valu + salu
    if (KIND == 47) BODY32("v_fma_f32 %0, %0, %1, %2\n\ts_add_u32 %3, %3, 7\n\ts_xor_b32 %3, %3, 5")
    if (KIND == 48) BODY32("v_readlane_b32 s22, %0, 3")
    if (KIND == 49) BODY32("v_and_b32 %0, 0x0f0f0f0f, %0")
    if (KIND == 50) BODY32("v_lshrrev_b32 %0, 4, %0")
    if (KIND == 51) BODY32("v_or_b32 %0, %0, %1")
    if (KIND == 52) BODY32("v_mul_u32_u24 %0, %0, %1")
    if (KIND == 53) BODY32("v_cvt_f32_i32 %0, %0")
    if (KIND == 54) BODY32("v_med3_f32 %0, %0, %1, %2")
    if (KIND == 55) BODY32("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
    // LDS under load (32 in flight then wait)
    if (KIND == 60) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 32; ++i) { asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"(la)); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    if (KIND == 61) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 16; ++i) { asm volatile("ds_read_b64 %0, %1" : "=v"(q[i]) : "v"(la)); } asm volatile("s_waitcnt lgkmcnt(0)"); _Pragma("unroll") for (int i = 0; i < 16; ++i) { asm volatile("ds_read_b64 %0, %1" : "=v"(q[i]) : "v"(la)); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    if (KIND == 62) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 32; ++i) { asm volatile("ds_add_u32 %0, %1" :: "v"(la), "v"(a[i])); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    if (KIND == 63) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 32; ++i) { asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(a[i]) : "v"(la), "v"(b)); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    if (KIND == 64) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 32; ++i) { asm volatile("ds_write_b32 %0, %1" :: "v"(la), "v"(a[i])); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    // one LDS op per 4 normal VALU ops (does LDS issue cost VALU issue time?)
    if (KIND == 65) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("ds_read_b32 %0, %4\n\tv_fma_f32 %1, %1, %5, %6\n\tv_fma_f32 %2, %2, %5, %6\n\tv_fma_f32 %3, %3, %5, %6" : "=v"(a[i]), "+v"(a[8 + i]), "+v"(a[16 + i]), "+v"(a[24 + i]) : "v"(la), "v"(b), "v"(c)); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    if (KIND == 66) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("ds_add_u32 %3, %0\n\tv_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5" : "+v"(a[i]), "+v"(a[8 + i]), "+v"(a[16 + i]) : "v"(la), "v"(b), "v"(c)); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) r ^= a[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) r ^= (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r ^ sg;
}

template <int KIND> void run(const char *name, double per_trip)
{
    const int ks[5] = {1, 2, 3, 4, 8};
    printf("%-44s", name);
    for (int ki = 0; ki < 5; ++ki) {
        const int kw = ks[ki];
        const int blocks = 256 * kw;
        const size_t lds_bytes = (size_t)(160 * 1024 / kw - 1024) & ~(size_t)1023;
        uint32_t *d;
        (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        const int iters = 4096 / kw;
        k<KIND><<<blocks, 256, lds_bytes>>>(d, 12345u, 8);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0); k<KIND><<<blocks, 256, lds_bytes>>>(d, 12345u, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr_simd = (double)iters * per_trip * kw;
        printf("  k=%d %5.2f", kw, (ms * 1e6 - 6000.0) / instr_simd);      // ~6 us of launch overhead removed
        (void)hipFree(d);
    }
    printf("\n");
    fflush(stdout);
}
int main()
{
    printf("ns per wave-instruction (or per group, as named) per SIMD, wall time\n");
    run<0>("v_add_u32 v,v", 32); run<1>("v_add_u32 s,v", 32); run<2>("v_and_b32 v,v", 32); run<3>("v_xor_b32 v,v", 32); run<51>("v_or_b32 v,v", 32);
    run<4>("v_lshlrev_b32 imm", 32); run<50>("v_lshrrev_b32 imm", 32); run<5>("v_sub_u32", 32); run<6>("v_min_u32", 32); run<7>("v_mov_b32", 32); run<49>("v_and_b32 literal", 32);
    run<8>("v_add_f32", 32); run<9>("v_mul_f32", 32); run<10>("v_fma_f32 v,v,v", 32); run<11>("v_max_f32", 32); run<54>("v_med3_f32", 32);
    run<12>("v_cmp_lt_f32 vcc", 32); run<13>("v_cmp_lt_f32 sgpr pair", 32); run<43>("v_cmp_eq_u32 vcc", 32);
    run<14>("v_cndmask vcc (in place)", 32); run<15>("v_cndmask sgpr pair", 32); run<16>("v_cndmask vcc dst!=src", 32);
    run<17>("v_cmp vcc; s_nop 1; v_cndmask (per group)", 32); run<18>("v_cmp sgpr; s_nop 1; v_cndmask (per group)", 32); run<55>("v_cmp vcc; v_cndmask no nop (per group)", 32);
    run<19>("v_addc_co_u32", 32); run<20>("v_add_u32_sdwa", 32); run<21>("v_pk_add_u16", 32); run<22>("v_pk_lshlrev_b16", 32);
    run<23>("v_lshl_add_u32", 32); run<24>("v_and_or_b32", 32); run<25>("v_bfi_b32", 32); run<26>("v_alignbit_b32", 32); run<27>("v_mbcnt_lo", 32);
    run<28>("v_cvt_i32_f32", 32); run<53>("v_cvt_f32_i32", 32); run<29>("v_cvt_f32_ubyte0", 32); run<30>("v_floor_f32", 32);
    run<31>("v_sad_u8", 32); run<32>("v_dot4_u32_u8", 32); run<33>("v_max3_u32", 32); run<52>("v_mul_u32_u24", 32);
    run<34>("v_sqrt_f32", 32); run<35>("v_rsq_f32", 32); run<36>("v_pk_fma_f32", 32); run<37>("v_pk_add_f32", 32); run<38>("v_lshlrev_b64", 32);
    run<39>("v_bcnt_u32_b32", 32); run<40>("v_perm_b32", 32); run<41>("v_add_u32_dpp row_shr", 32); run<42>("v_mov_b32_dpp row_shr", 32); run<48>("v_readlane_b32", 32);
    run<44>("v_sub_f32; v_mul_f32 dependent (per pair)", 32); run<45>("v_add_u32 + s_add (per pair)", 32); run<46>("v_fma + s_add (per pair)", 32); run<47>("v_fma + 2 salu (per triple)", 32);
    run<60>("ds_read_b32 x32", 32); run<61>("ds_read_b64 x16 x2", 32); run<62>("ds_add_u32 x32", 32); run<63>("ds_add_rtn_u32 x32", 32); run<64>("ds_write_b32 x32", 32);
    run<65>("ds_read_b32 + 3 v_fma (per group)", 8); run<66>("ds_add_u32 + 3 v_fma (per group)", 8);
    return 0;
}
