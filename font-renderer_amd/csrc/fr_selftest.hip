// fr_selftest.hip — exhaustive on-device proof obligations for arithmetic shortcuts.
//
// div_by_int (fr_device.hpp) replaces the IEEE division  t = num / d  of
// /root/reference/src/tools/render_glyph.zig:51,60-61 by a reciprocal multiply plus Markstein's
// FMA correction.  That is only admissible if it returns the SAME binary32 for every input the
// render kernel can feed it.  d is an integer (a = p0y - 2 p1y + p2y, or p2y - p0y) with
// |d| <= 4*32768 = 2^17.  This kernel checks, for every integer d in [d_lo, d_hi] and EVERY one
// of the 2^23 significands x in [1, 2) — plus the same significands at two other binades and
// with both signs of d — that div_by_int(x, d, RN(1/d)) == x / d bit for bit (hipcc's default
// correctly-rounded division).  Scaling x by 2^k is exact for q, r and q', so one binade
// represents them all while every intermediate stays normal.
#include "../../include/fr_raster.h"
#include "fr_device.hpp"

namespace fr {

__global__ __launch_bounds__(256) void selftest_div_kernel(uint32_t d_lo, uint32_t n_d,
                                                           unsigned long long *mismatches,
                                                           uint32_t *first_bad)
{
    const uint32_t d_idx = blockIdx.x / 32u, part = blockIdx.x % 32u;
    if (d_idx >= n_d) return;
    const float d = (float)(d_lo + d_idx);
    const float rd = 1.0f / d, rdn = 1.0f / (-d);
    uint32_t bad = 0, bad_x = 0;
    // 2^23 significands split over 32 blocks x 256 threads
    for (uint32_t m = part * 256u + threadIdx.x; m < (1u << 23); m += 32u * 256u) {
        const float x = __uint_as_float(0x3f800000u | m);               // [1, 2)
        const float xs = __uint_as_float(0x2f800000u | m);              // [2^-32, 2^-31)
        const float xl = __uint_as_float(0x4f800000u | m);              // [2^32, 2^33)
        const bool ok = (__float_as_uint(div_by_int(x, d, rd)) == __float_as_uint(x / d)) &&
                        (__float_as_uint(div_by_int(x, -d, rdn)) == __float_as_uint(x / -d)) &&
                        (__float_as_uint(div_by_int(-xs, d, rd)) == __float_as_uint(-xs / d)) &&
                        (__float_as_uint(div_by_int(xl, d, rd)) == __float_as_uint(xl / d));
        if (!ok) { ++bad; bad_x = 0x3f800000u | m; }
    }
    if (bad) {
        atomicAdd(mismatches, (unsigned long long)bad);
        atomicMax(&first_bad[0], d_lo + d_idx);
        first_bad[1] = bad_x;
    }
}

// sqrt_rn (fr_device.hpp) == the correctly rounded square root for every binary32 whose exponent
// field lies in [e_lo, e_hi] (positive, normal): one block row per exponent, all 2^23 significands.
__global__ __launch_bounds__(256) void selftest_sqrt_kernel(uint32_t e_lo, unsigned long long *mismatches, uint32_t *first_bad)
{
    const uint32_t e = e_lo + blockIdx.x / 32u, part = blockIdx.x % 32u;
    uint32_t bad = 0, bad_x = 0;
    for (uint32_t m = part * 256u + threadIdx.x; m < (1u << 23); m += 32u * 256u) {
        const uint32_t bits = (e << 23) | m;
        const float x = __uint_as_float(bits);
        if (__float_as_uint(sqrt_rn(x)) != __float_as_uint(__builtin_sqrtf(x))) { ++bad; bad_x = bits; }
    }
    if (bad) {
        atomicAdd(mismatches, (unsigned long long)bad);
        first_bad[0] = bad_x;
    }
}

}  // namespace fr

extern "C" int fr_selftest_sqrt(uint64_t *mismatches, uint32_t *bad_x_bits)
{
    if (!mismatches) return FR_E_INVALID;
    unsigned long long *d_m = nullptr;
    uint32_t *d_b = nullptr;
    if (hipMalloc(&d_m, 8) != hipSuccess || hipMalloc(&d_b, 8) != hipSuccess) return FR_E_HIP;
    (void)hipMemset(d_m, 0, 8);
    (void)hipMemset(d_b, 0, 8);
    // exponent fields 97 .. 192: [2^-30, 2^66) covers every delta the path can form (and then some)
    const uint32_t e_lo = 97u, e_hi = 192u;
    hipLaunchKernelGGL(fr::selftest_sqrt_kernel, dim3((e_hi - e_lo + 1u) * 32u), dim3(256), 0, 0, e_lo, d_m, d_b);
    // the two special arguments: 0 -> 0, negative -> NaN are checked on the host side of the test
    unsigned long long m = 0;
    uint32_t b = 0;
    hipError_t e = hipMemcpy(&m, d_m, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&b, d_b, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_m);
    (void)hipFree(d_b);
    if (e != hipSuccess) return FR_E_HIP;
    *mismatches = m;
    if (bad_x_bits) *bad_x_bits = b;
    return FR_OK;
}

extern "C" int fr_selftest_division(uint32_t d_lo, uint32_t d_hi, uint64_t *mismatches,
                                    uint32_t *bad_divisor, uint32_t *bad_x_bits)
{
    if (!mismatches || d_lo < 1 || d_hi < d_lo || d_hi > (1u << 17)) return FR_E_INVALID;
    unsigned long long *d_m = nullptr;
    uint32_t *d_b = nullptr;
    if (hipMalloc(&d_m, 8) != hipSuccess || hipMalloc(&d_b, 8) != hipSuccess) return FR_E_HIP;
    (void)hipMemset(d_m, 0, 8);
    (void)hipMemset(d_b, 0, 8);
    const uint32_t n_d = d_hi - d_lo + 1;
    for (uint32_t off = 0; off < n_d; off += 4096u) {              // <= 131072 blocks per launch
        const uint32_t n = n_d - off < 4096u ? n_d - off : 4096u;
        hipLaunchKernelGGL(fr::selftest_div_kernel, dim3(n * 32u), dim3(256), 0, 0, d_lo + off, n, d_m, d_b);
    }
    unsigned long long m = 0;
    uint32_t b[2] = {0, 0};
    hipError_t e = hipMemcpy(&m, d_m, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(b, d_b, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_m);
    (void)hipFree(d_b);
    if (e != hipSuccess) return FR_E_HIP;
    *mismatches = m;
    if (bad_divisor) *bad_divisor = b[0];
    if (bad_x_bits) *bad_x_bits = b[1];
    return FR_OK;
}
