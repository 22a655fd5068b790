// fr_device.hpp — shared device-side definitions for the gfx950 glyph rasterizer.
//
// Arithmetic contract: the reference's glyphWindingAt
// (/root/reference/src/tools/render_glyph.zig:35-73) is strict IEEE-754 binary32
// with no fused multiply-add.  Every translation unit that includes this header is
// compiled with -ffp-contract=off and keeps hipcc's default correctly-rounded
// f32 divide / sqrt; the pragma below makes the first part independent of flags.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace fr {

// ---------------------------------------------------------------------------
// Root records.  One quadratic segment (p0,p1,p2) of a contour yields up to two
// records (its t+ and t- root; render_glyph.zig:60-63) or one (the a == 0 branch,
// :49-57).  A record keeps exactly the f32 constants the reference recomputes per
// pixel, plus [lo, hi]: the closed interval of ray heights cy for which the
// reference ACCEPTS this root (delta >= 0 and 0 <= t < 1, :52,:59,:64).  The
// interval is exact, not an estimate — see fr_prepare.hip.
// ---------------------------------------------------------------------------
// 48 B = three dwordx4:
//   quadratic: lo, hi, a, B = p0y - p1y, c1 = p1y*p1y, c2 = p0y*p2y, Ax, Bx, p0x, flags
//   linear   : lo, hi, 0, p0y,           den = p2y-p0y, 0,            Ax, Bx, p0x, flags
//   Ax = (p0x - 2*p1x) + p2x, Bx = 2*(p1x - p0x)               (:53, :65)
struct __attribute__((aligned(16))) Rec {
    float lo, hi, a, b, c1, c2, ax, bx, p0x;
    uint32_t flags;
    float rden;          // RN(1 / d), d = the divisor of t: a (quadratic) or p2y - p0y (linear)
    uint32_t sqsign;     // 0x80000000 for the t- root (XORed into sqrt(delta): B - sqrt == B + (-sqrt)), else 0
};

// x / d for an INTEGER divisor |d| <= 2^17 with rd = RN(1/d): Markstein's correction
//   q = RN(x*rd); r = x - q*d (exact in one FMA); q' = RN(q + r*rd)
// returns the correctly rounded quotient — bit-identical to IEEE division — for every binary32
// x whose quotient and residual stay normal.  Not taken on trust: fr_selftest_division
// (fr_selftest.hip) compares it with hipcc's IEEE `/` for ALL 2^23 significands x EVERY integer
// divisor 1..2^17 on the GPU; the exponent of x is irrelevant (scaling by 2^k is exact) as long
// as nothing leaves the normal range, which fr_plan_create guarantees by bounding `scale`
// (DESIGN.md §3.4).  The FMAs here implement a division; they are not a contraction of the
// reference's expression.
__device__ __forceinline__ float div_by_int(float x, float d, float rd)
{
    const float q = x * rd;
    const float r = __builtin_fmaf(-q, d, x);
    return __builtin_fmaf(r, rd, q);
}
// Correctly rounded sqrt for the arguments the path can produce: delta of render_glyph.zig:58 is
// 0, negative, or at least 2^-24 in magnitude (c1, c2 are integers, |a| >= 1, and a non-zero ray
// height is at least 2^-23: DESIGN.md §3), never a denormal — so the 2^32 pre-scaling of the
// general lowering is dead weight.  v_sqrt_f32 is within 1 ulp; the two FMA residuals pick the
// neighbour exactly as the compiler's own lowering does.  Negative -> NaN, 0 -> 0.  Not taken on
// trust: fr_selftest_sqrt compares it with __builtin_sqrtf for EVERY binary32 in [2^-30, 2^66).
__device__ __forceinline__ float sqrt_rn(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
    const float sp = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
    const float rm = __builtin_fmaf(-sm, s, x);          // x - (s - ulp) * s
    const float rp = __builtin_fmaf(-sp, s, x);          // x - (s + ulp) * s
    float r = (rm <= 0.0f) ? sm : s;
    r = (rp > 0.0f) ? sp : r;
    return r;
}
enum : uint32_t {
    REC_LINEAR = 0x80000000u,   // a == 0 branch (the sign bit: one signed compare tests it)
    REC_NEG_ROOT = 2u,   // t- = (B - sqrt(delta)) / a ; otherwise t+
    REC_LIN_MINUS = 4u,  // linear branch adds -1 (p0y < p2y, :55) ; otherwise +1 ...
    REC_LIN_PLUS = 2u    // ... and then carries this bit too: (flags & 2) is the crossing's step code (fr_render.hip)
};

struct Job {   // == fr_job (include/fr_raster.h)
    uint32_t glyph;
    int32_t min_x, max_y;
    uint32_t w, h;
    uint32_t out_x, out_y;
    float scale;
};

// kernel arguments of render_kernel (fr_render.hip), filled by fr_api.hip
struct RenderArgs {
    const Job *jobs;
    const uint32_t *glyph_seg_start;   // record slice of glyph g starts at 2*glyph_seg_start[g]
    const uint32_t *job_seg;           // [n_jobs][2]: glyph_seg_start[g] and the segment count of job j's glyph g
    const uint32_t *glyph_rec_count;
    const Rec *recs;
    const int16_t *pts;                // glyph points / per-segment p0 index: the render kernel builds its
    const uint32_t *seg_p0;            // glyph's records itself (in LDS) when `fused` is set
    const int16_t *seg_pts;            // the same control points laid out per segment (6 x i16 each): cov4_kernel reads
                                       // its segment with one load instead of two dependent ones
    const uint32_t *job_bits;          // win1_kernel's sign-bit mode / sdf_kernel: first word of each job's bit plane, or
                                       // 0xffffffff for a job whose sign is the byte the general kernel left in the output
    const uint32_t *bits;              // sdf_kernel: the bit planes (win1_kernel's sign-bit mode writes them through `out`)
    uint32_t fused;
    uint32_t uniform;                  // every job: w a multiple of strip_w, h a multiple of the wave band (64/n rows)
    void *out;
    uint64_t out_stride;               // elements
    uint32_t n_jobs, bands, strips, strip_w, kmax;
    uint32_t bands_per_wg, band_groups;   // a workgroup walks bands_per_wg consecutive bands of its cell
    int32_t phase_center;
    uint32_t lds_pad;
    uint32_t nwin_log, lds_region, lds_rec_bytes, lds_wave_bytes, lds_tail;   // filled by launch_render (LDS plan)
};

// order-preserving map binary32 -> u32 (total order, -0 < +0)
__host__ __device__ inline uint32_t f2key(float f)
{
    uint32_t b;
    __builtin_memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float key2f(uint32_t k)
{
    uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}

// t of a record at ray height cy — the reference's operation order, one rounding
// per operation:  delta = ((cy*a) + c1) - c2 (:58);  t = (B +/- sqrt(delta)) / a (:60-61)
//                 linear: t = (cy - p0y) / (p2y - p0y) (:51)
__device__ __forceinline__ float rec_t_quad(const Rec &r, float cy)
{
    float delta = cy * r.a + r.c1 - r.c2;
    float sq = __builtin_sqrtf(delta);
    float num = (r.flags & REC_NEG_ROOT) ? (r.b - sq) : (r.b + sq);
    return num / r.a;
}
__device__ __forceinline__ float rec_t_lin(const Rec &r, float cy)
{
    return (cy - r.b) / r.c1;
}
// the reference's decision for one record at ray height cy (:49-69): is the root accepted,
// where does it cross (xx) and which sign does it add.  [lo, hi] only brackets the accepted
// set, so the three rejection tests (:52, :59, :64) are applied here exactly as written.
__device__ __forceinline__ bool rec_cross(const Rec &r, float cy, float &xx, int &sgn)
{
    float t;
    bool ok;
    if (r.flags & REC_LINEAR) {
        t = rec_t_lin(r, cy);
        ok = !(t < 0.0f || t >= 1.0f);                                  // :52
        sgn = (r.flags & REC_LIN_MINUS) ? -1 : 1;
    } else {
        const float delta = cy * r.a + r.c1 - r.c2;                     // :58
        t = rec_t_quad(r, cy);
        ok = !(delta < 0.0f) && !(t < 0.0f || t >= 1.0f);               // :59, :64
        float dy = r.a * t + (-r.b);           // a*t + (p1y - p0y), :67
        sgn = (dy > 0.0f) ? -1 : 1;            // :68
    }
    xx = (r.ax * t + r.bx) * t + r.p0x;        // :53 / :65
    return ok;
}

// direct sum over a glyph's records — the reference's own loop shape (render_glyph.zig:37-71):
// used for the sign of the SDF (the render kernel's over-full rows use rec_cross directly)
__device__ inline int brute_winding(const Rec *__restrict__ recs, uint32_t n, float cx, float cy)
{
    int w = 0;
    for (uint32_t c = 0; c < n; ++c) {
        const Rec r = recs[c];
        if (cy >= r.lo && cy <= r.hi) {
            float xx; int sgn;
            if (rec_cross(r, cy, xx, sgn) && !(xx < cx)) w += sgn;
        }
    }
    return w;
}

// sub-sample offset (k + phase)/n — exact in binary32 for n in {1,2,4}
__device__ __forceinline__ float sub_off(int k, int n, int phase_center)
{
    return ((float)k + (phase_center ? 0.5f : 0.0f)) / (float)n;
}

}  // namespace fr
