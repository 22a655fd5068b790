// How often does the raw v_sqrt_f32 / v_rcp_f32-based division of gfx950 differ from the correctly rounded result?
// (sqrt_rn and div_by_int of fr_device.hpp pay 7 + 2 instructions per crossing for the correction.)
// usage: sqrt_exact  — every binary32 in [2^-30, 2^66)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned long long *bad, unsigned long long *bad_lo, unsigned long long *bad_hi)
{
    const uint32_t lo = 0x30800000u, hi = 0x60800000u;
    unsigned long long n = 0, nl = 0, nh = 0;
    for (uint64_t b = lo + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < hi; b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __builtin_bit_cast(float, (uint32_t)b);
        const float raw = __builtin_amdgcn_sqrtf(x), ref = __builtin_sqrtf(x);
        if (raw != ref) { ++n; if (raw < ref) ++nl; else ++nh; }
    }
    if (n) { atomicAdd(bad, n); atomicAdd(bad_lo, nl); atomicAdd(bad_hi, nh); }
}
int main()
{
    unsigned long long *d, h[3] = {0, 0, 0};
    hipMalloc(&d, 24); hipMemcpy(d, h, 24, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, d + 1, d + 2);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("v_sqrt_f32 vs correctly rounded sqrt over [2^-30, 2^66): %llu of %llu inputs differ (%llu low, %llu high)\n", h[0],
           (unsigned long long)(0x60800000u - 0x30800000u), h[1], h[2]);
    return 0;
}
