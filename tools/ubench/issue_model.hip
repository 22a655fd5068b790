// issue_model.hip — how gfx950 issues the instruction kinds the render kernel is made of, as a
// function of waves per SIMD (1, 2, 3, 4, 8).  Answers: is a wave64 VALU op 2 or 4 cycles of a SIMD?
// do scalar ops of one wave overlap vector ops of another at 3 waves/SIMD?  what do packed-f32, DPP,
// v_perm, v_bcnt, 64-bit shifts, transcendental and LDS atomic instructions cost?
// Output: cycles per wave-instruction per SIMD (s_memtime ticks of the slowest wave / instructions
// issued on that SIMD), one line per kind.  Build: hipcc --offload-arch=gfx950 -O2 issue_model.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP8(S) S S S S S S S S
// independent: 8 registers, each its own chain
#define BODY_I(ASM) \
    for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "s"(sg)); } }
// dependent: one register, one chain
#define BODY_D(ASM) \
    for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile(ASM : "+v"(a[0]) : "v"(b), "v"(c), "s"(sg)); } }
// 64-bit independent
#define BODY_I64(ASM) \
    for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile(ASM : "+v"(q[i]) : "v"(b), "v"(c), "s"(sg)); } }

extern __shared__ uint32_t lds[];

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *cyc, uint32_t seed, int iters)
{
    uint32_t a[8]; uint32_t b = seed + threadIdx.x, c = seed * 3 + threadIdx.x; uint32_t sg = seed * 7;
    unsigned long long q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed * (i + 1) + threadIdx.x; q[i] = 0x0123456789abcdefull * (a[i] | 1u); }
    uint32_t la = (threadIdx.x * 8u) & 0x3ffu;          // LDS byte address (8-byte aligned, conflict-free for b64)
    lds[threadIdx.x] = seed; lds[threadIdx.x + 256] = seed;
    __syncthreads();
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(b), "v"(c) : "vcc");
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (KIND == 0) BODY_I("v_add_u32 %0, %0, %1")
    if (KIND == 1) BODY_D("v_add_u32 %0, %0, %1")
    if (KIND == 2) BODY_I("v_fma_f32 %0, %0, %3, %1")
    if (KIND == 3) BODY_D("v_fma_f32 %0, %0, %3, %1")
    if (KIND == 4) BODY_I64("v_pk_fma_f32 %0, %0, %0, %0")
    if (KIND == 5) BODY_I64("v_pk_mul_f32 %0, %0, %0")
    if (KIND == 6) BODY_I64("v_pk_add_f32 %0, %0, %0")
    if (KIND == 7) BODY_I("v_sqrt_f32 %0, %0")
    if (KIND == 8) BODY_I("v_rcp_f32 %0, %0")
    if (KIND == 9) BODY_I("v_perm_b32 %0, %0, %1, %2")
    if (KIND == 10) BODY_I("v_bcnt_u32_b32 %0, %0, %1")
    if (KIND == 11) BODY_I("v_bfe_u32 %0, %0, 3, 7")
    if (KIND == 12) BODY_I("v_cndmask_b32 %0, %0, %1, vcc")
    if (KIND == 13) BODY_I64("v_lshrrev_b64 %0, 3, %0")
    if (KIND == 14) BODY_I64("v_lshlrev_b64 %0, %1, %0")
    if (KIND == 15) BODY_I("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
    if (KIND == 16) BODY_D("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
    if (KIND == 17) BODY_I("v_mul_lo_u32 %0, %0, %1")
    if (KIND == 18) BODY_I("v_mad_u32_u24 %0, %0, %1, %2")
    if (KIND == 19) BODY_I("v_cvt_f32_i32 %0, %0")
    if (KIND == 20) BODY_I("v_pk_min_u16 %0, %0, %1")
    if (KIND == 21) BODY_I("v_and_b32 %0, 0x0f0f0f0f, %0")
    if (KIND == 22) BODY_I("v_xor_b32 %0, %3, %0")
    if (KIND == 23) BODY_I("v_med3_f32 %0, %0, %1, %2")
    if (KIND == 24) BODY_I("v_add3_u32 %0, %0, %1, %2")
    if (KIND == 25) BODY_I("v_lshl_or_b32 %0, %0, 3, %1")
    // VALU + SALU 1:1 (does the scalar op of one wave hide under the vector op of another?)
    if (KIND == 30) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 7" : "+v"(a[i]), "+s"(sg) : "v"(b) : "scc"); } } }
    // VALU + 2 SALU
    if (KIND == 31) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 7\n\ts_xor_b32 %1, %1, 5" : "+v"(a[i]), "+s"(sg) : "v"(b) : "scc"); } } }
    // SALU only
    if (KIND == 32) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("s_add_u32 %0, %0, 7" : "+s"(sg) :: "scc"); } } }
    // v_cmp -> sgpr pair + s_and (ballot-like)
    if (KIND == 33) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "s20", "s21"); } } }
    // v_readlane (VALU -> SGPR)
    if (KIND == 34) BODY_I("v_readlane_b32 s20, %0, 3")
    if (KIND == 35) BODY_I("v_readfirstlane_b32 s20, %0")
    // LDS: independent reads, waited once per 8
    if (KIND == 40) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("ds_read_b64 %0, %1" : "=v"(q[i]) : "v"(la)); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    if (KIND == 41) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("ds_read_b128 %0, %1" : "=v"(*(reinterpret_cast<__uint128_t *>(&q[i & 6]))) : "v"((la * 2) & 0x7ff)); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    // LDS atomics without return
    if (KIND == 42) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("ds_xor_b64 %0, %1" :: "v"(la), "v"(q[i])); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    // LDS atomics with return, all 8 in flight
    if (KIND == 43) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(a[i]) : "v"(la), "v"(b)); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    // LDS atomic with return, waited each (latency)
    if (KIND == 44) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(a[i]) : "v"(la), "v"(b)); } } }
    if (KIND == 45) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a[i]) : "v"(la)); } } }
    if (KIND == 46) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("ds_write_b16 %0, %1" :: "v"(la), "v"(a[i])); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    if (KIND == 47) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("ds_write_b64 %0, %1" :: "v"(la), "v"(q[i])); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    if (KIND == 48) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("ds_or_b32 %0, %1" :: "v"(la), "v"(a[i])); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    // mixed: 4 VALU + 1 LDS read in flight (LDS issue hidden under VALU?)
    if (KIND == 49) { for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { asm volatile("ds_read_b64 %0, %2\n\tv_add_u32 %1, %1, %3\n\tv_add_u32 %1, %1, %3\n\tv_add_u32 %1, %1, %3\n\tv_add_u32 %1, %1, %3" : "=v"(q[i]), "+v"(a[i]) : "v"(la), "v"(b)); } asm volatile("s_waitcnt lgkmcnt(0)"); } }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r ^ sg;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static int g_clock_khz = 0;
template <int KIND> void run(const char *name, double per8)
{
    const int ks[5] = {1, 2, 3, 4, 8};
    printf("%-34s", name);
    for (int ki = 0; ki < 5; ++ki) {
        const int kw = ks[ki];                       // workgroups (of 4 waves) per CU = waves per SIMD
        const int blocks = 256 * kw;
        const size_t lds_bytes = (size_t)(160 * 1024 / kw - 1024) & ~(size_t)1023;   // exactly kw workgroups fit a CU
        uint32_t *d; unsigned long long *dc;
        hipMalloc(&d, (size_t)blocks * 256 * 4); hipMalloc(&dc, (size_t)blocks * 4 * 8);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        const int iters = 1024;
        k<KIND><<<blocks, 256, lds_bytes>>>(d, dc, 12345u, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); k<KIND><<<blocks, 256, lds_bytes>>>(d, dc, 12345u, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)blocks * 4);
        hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        const double instr_wave = (double)iters * per8;          // instructions one wave issued
        // cycles per instruction per SIMD: a wave's ticks / (its instructions x waves sharing the SIMD)
        printf("  k=%d %6.2f (w %5.2fns)", kw, med / (instr_wave * kw), ms * 1e6 / (instr_wave * kw));
        hipFree(d); hipFree(dc);
    }
    printf("\n");
}
int main()
{
    printf("cycles (s_memtime ticks, 100 MHz ref x ? — compare columns) per wave-instruction per SIMD; (w: wall ns per wave-instr per SIMD)\n");
    run<0>("v_add_u32 indep", 8); run<1>("v_add_u32 dependent", 8);
    run<2>("v_fma_f32 indep", 8); run<3>("v_fma_f32 dependent", 8);
    run<4>("v_pk_fma_f32 indep", 8); run<5>("v_pk_mul_f32 indep", 8); run<6>("v_pk_add_f32 indep", 8);
    run<7>("v_sqrt_f32", 8); run<8>("v_rcp_f32", 8);
    run<9>("v_perm_b32", 8); run<10>("v_bcnt_u32_b32", 8); run<11>("v_bfe_u32", 8); run<12>("v_cndmask_b32", 8);
    run<13>("v_lshrrev_b64 imm", 8); run<14>("v_lshlrev_b64 vgpr", 8);
    run<15>("v_add_u32_dpp row_shr indep", 8); run<16>("v_add_u32_dpp row_shr dependent", 8);
    run<17>("v_mul_lo_u32", 8); run<18>("v_mad_u32_u24", 8); run<19>("v_cvt_f32_i32", 8);
    run<20>("v_pk_min_u16", 8); run<21>("v_and_b32 literal", 8); run<22>("v_xor_b32 sgpr", 8);
    run<23>("v_med3_f32", 8); run<24>("v_add3_u32", 8); run<25>("v_lshl_or_b32", 8);
    run<30>("v_add + s_add (per pair)", 8); run<31>("v_add + 2 s_op (per triple)", 8); run<32>("s_add_u32 only", 8);
    run<33>("v_cmp->sgpr + v_add (per pair)", 8); run<34>("v_readlane_b32", 8); run<35>("v_readfirstlane_b32", 8);
    run<40>("ds_read_b64 x8 then wait", 8); run<41>("ds_read_b128 x8 then wait", 8);
    run<42>("ds_xor_b64 x8 then wait", 8); run<43>("ds_add_rtn_u32 x8 then wait", 8);
    run<44>("ds_add_rtn_u32 waited each", 8); run<45>("ds_read_b32 waited each", 8);
    run<46>("ds_write_b16 x8 then wait", 8); run<47>("ds_write_b64 x8 then wait", 8); run<48>("ds_or_b32 x8 then wait", 8);
    run<49>("ds_read_b64 + 4 v_add (per 5)", 8);
    return 0;
}
