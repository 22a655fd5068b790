// VALU issue-rate microbenchmark (gfx950): cycles per wave64 instruction per SIMD for the
// instruction kinds the render kernel is made of.  8 waves/SIMD, 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ __launch_bounds__(512) void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed * (i + 1) + threadIdx.x; f[i] = (float)a[i] * 1e-9f; }
    unsigned long long m = 0x123456789abcdefull * (threadIdx.x + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);
            if (KIND == 1) a[i] = a[i] + seed;
            if (KIND == 2) a[i] = a[i] & (seed | 0xf0f0f0f0u);
            if (KIND == 3) a[i] = min(a[i], a[(i + 1) & 7] ^ seed);
            if (KIND == 4) a[i] = (a[i] > seed) ? a[(i + 1) & 7] : a[i] + 1;      // cmp + cndmask
            if (KIND == 5) { m = (m >> (a[i] & 31)) + i; a[i] ^= (uint32_t)m; }    // 64-bit shift
            if (KIND == 6) a[i] = __builtin_amdgcn_readlane(a[i], it & 63) + a[i]; // readlane + add
            if (KIND == 7) f[i] = f[i] * 1.0001f;
            if (KIND == 8) a[i] = __builtin_amdgcn_perm(a[i], seed, 0x05010400u);
            if (KIND == 9) a[i] = (a[i] << 3) | (a[i] >> 7);
        }
    }
    uint32_t r = (uint32_t)m;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ __float_as_uint(f[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND> void run(const char *name, int per_iter)
{
    uint32_t *d; hipMalloc(&d, 256 * 4 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    k<KIND><<<256 * 4, 512>>>(d, 12345u, 16);
    hipEventRecord(e0); k<KIND><<<256 * 4, 512>>>(d, 12345u, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 4 WGs/CU x 8 waves = 32 waves/CU = 8 per SIMD; instructions per SIMD = 8 waves * iters * 8 * per_iter
    double instr = 8.0 * iters * 8 * per_iter;
    printf("%-28s %.3f ms  -> %.2f ns per wave-instr per SIMD (x clock GHz = cycles)\n", name, ms, ms * 1e6 / instr);
    hipFree(d);
}
int main()
{
    run<0>("v_fma_f32", 1); run<7>("v_mul_f32", 1); run<1>("v_add_u32", 1); run<2>("v_and_b32 (+or)", 1);
    run<3>("v_xor + v_min_u32", 2); run<4>("v_cmp + v_cndmask + v_add", 3); run<5>("v_lshrrev_b64 + adds", 4);
    run<6>("v_readlane + v_add", 2); run<8>("v_perm_b32", 1); run<9>("shl+shr+or (or v_alignbit)", 3);
    return 0;
}
