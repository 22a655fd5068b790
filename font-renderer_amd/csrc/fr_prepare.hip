// fr_prepare.hip — per-segment precompute: root records with EXACT acceptance intervals.
//
// Why an interval is exact.  For one root of one segment the reference decides
// "accept" from (render_glyph.zig:58-64)
//        delta = fl(fl(fl(cy*a) + c1) - c2);   reject if delta < 0
//        t     = fl(fl(B +/- sqrt(delta)) / a); reject if t < 0 or t >= 1
// with a, B, c1, c2 constants of the segment.  Every step is a correctly rounded
// IEEE operation with one varying operand, hence a monotone function of it; so
// delta(cy) is monotone (direction = sign a), sqrt is non-decreasing, and
// t+(cy) is non-decreasing / t-(cy) non-increasing for either sign of a.  The
// three rejection tests are therefore each a one-sided cut of the cy axis, and the
// accepted set is a closed interval [lo, hi] of binary32 values.  We locate its two
// ends by searching the ordered bit patterns of ALL finite floats (exponential search out
// of analytic hints, then bisection), evaluating the reference's own expression
// (rec_t_quad / rec_t_lin) at each probe — no error analysis, no tolerance: the hints
// only decide where the search starts, never what it returns.  The same holds for the a == 0 branch (:51-52).
// With [lo, hi] known the render kernel tests  lo <= cy <= hi  (two compares) instead
// of solving the quadratic for every (row, segment), and rows outside it are culled
// with a bit-exact guarantee.
#include "fr_device.hpp"

namespace fr {

// 0 = cy lies below the accepted interval, 1 = accepted, 2 = above
__device__ __forceinline__ int classify(const Rec &r, float cy)
{
    if (r.flags & REC_LINEAR) {
        float t = rec_t_lin(r, cy);
        bool up = r.c1 > 0.0f;                         // den > 0: t non-decreasing in cy
        if (t < 0.0f) return up ? 0 : 2;               // :52
        if (t >= 1.0f) return up ? 2 : 0;
        return 1;
    }
    float delta = cy * r.a + r.c1 - r.c2;              // :58
    if (delta < 0.0f) return (r.a > 0.0f) ? 0 : 2;     // :59
    float t = rec_t_quad(r, cy);
    bool neg = (r.flags & REC_NEG_ROOT) != 0;          // t- is non-increasing in cy
    if (t < 0.0f) return neg ? 2 : 0;                  // :64
    if (t >= 1.0f) return neg ? 0 : 2;
    return 1;
}

// Smallest key k in [L, H] such that class(k) >= T, given the invariant
//   every key <  L has class <  T,   every key >= H has class >= T      (class is monotone).
// Two-sided exponential search, then bisection: the cut almost always sits a few ulps
// from one end of the bracket (the ends are the segment's p0y / p2y / vertex height),
// so this takes ~2*log2(distance) probes instead of 32.
__device__ inline uint32_t first_at_least(const Rec &r, int T, uint32_t L, uint32_t H)
{
    uint32_t step = 1u;
    while (step < (1u << 30) && H - L > 2u * step) {
        const uint32_t pl = L + step - 1u;                  // probe near the low end
        if (classify(r, key2f(pl)) >= T) { H = pl; break; }
        L = pl + 1u;
        const uint32_t ph = H - step;                       // probe near the high end (ph >= L here)
        if (classify(r, key2f(ph)) < T) { L = ph + 1u; break; }
        H = ph;
        step <<= 1;
    }
    while (L < H) {
        const uint32_t mid = L + ((H - L) >> 1);
        if (classify(r, key2f(mid)) >= T) H = mid; else L = mid + 1u;
    }
    return L;
}

__device__ inline bool accept_interval(const Rec &r, float p0y, float p2y, float &lo, float &hi)
{
    const uint32_t kmin = f2key(-3.402823466e+38f), kmax = f2key(3.402823466e+38f);
    // bracket both cuts with the three heights where they can sit in exact arithmetic:
    // t = 0 at cy = p0y, t = 1 at cy = p2y, delta = 0 at the parabola's vertex height
    // y(t) = p0y - 2 B t + a t^2, t_v = B / a
    float cand[3];
    int ncand = 2;
    cand[0] = p0y; cand[1] = p2y;
    if (!(r.flags & REC_LINEAR)) { cand[2] = p0y - (r.b * r.b) / r.a; ncand = 3; }
    uint32_t L1 = kmin, H1 = kmax + 1u, L2 = kmin, H2 = kmax + 1u;
    for (int i = 0; i < ncand; ++i) {
        float c = cand[i];
        if (!(c >= -3.402823466e+38f && c <= 3.402823466e+38f)) continue;   // hints only
        const uint32_t k = f2key(c);
        const int cl = classify(r, c);
        if (cl >= 1) { if (k < H1) H1 = k; } else { if (k + 1u > L1) L1 = k + 1u; }
        if (cl >= 2) { if (k < H2) H2 = k; } else { if (k + 1u > L2) L2 = k + 1u; }
    }
    const uint32_t first = first_at_least(r, 1, L1, H1);    // first accepted key
    if (L2 < first) L2 = first;                             // class 2 cannot start below it
    if (H2 < L2) H2 = L2;
    const uint32_t a = first_at_least(r, 2, L2, H2);        // first key above the interval
    if (a == kmin || first > a - 1u) return false;
    lo = key2f(first);
    hi = key2f(a - 1u);
    return true;
}

// One wave64 per glyph.  Lane l builds candidate record c = base + l, where
// candidate 2s / 2s+1 are the t+ / t- roots of segment s (2s alone for a == 0);
// survivors are compacted with a ballot + prefix popcount into the glyph's slice
// [2*seg_start[g], ...) of the record arrays.
__global__ __launch_bounds__(64) void prepare_kernel(const int16_t *__restrict__ pts,
                                                     const uint32_t *__restrict__ seg_p0,
                                                     const uint32_t *__restrict__ glyph_seg_start,
                                                     uint32_t n_glyphs,
                                                     Rec *__restrict__ out_recs,
                                                     uint32_t *__restrict__ glyph_rec_count)
{
    const uint32_t g = blockIdx.x;
    if (g >= n_glyphs) return;
    const uint32_t s0 = glyph_seg_start[g], s1 = glyph_seg_start[g + 1];
    const uint32_t n_cand = 2u * (s1 - s0);
    const uint32_t lane = threadIdx.x;
    const size_t out_base = 2u * (size_t)s0;
    uint32_t n_out = 0;
    for (uint32_t base = 0; base < n_cand; base += 64u) {
        const uint32_t c = base + lane;
        bool valid = false;
        Rec r;
        r.rden = 0.0f; r.pad1 = 0;
        float lo = 0.f, hi = 0.f;
        if (c < n_cand) {
            const uint32_t s = s0 + (c >> 1);
            const uint32_t root = c & 1u;
            const int16_t *p = pts + 2u * (size_t)seg_p0[s];
            // i16 -> f32 (render_glyph.zig:43-45)
            const float p0x = (float)p[0], p0y = (float)p[1];
            const float p1x = (float)p[2], p1y = (float)p[3];
            const float p2x = (float)p[4], p2y = (float)p[5];
            const float a = p0y - 2 * p1y + p2y;                 // :48
            r.ax = p0x - 2 * p1x + p2x;                          // :53/:65
            r.bx = 2 * (p1x - p0x);
            r.p0x = p0x;
            if (a == 0.0f) {                                     // :49
                if (root == 0u && p2y != p0y) {                  // :50
                    r.a = 0.0f; r.b = p0y; r.c1 = p2y - p0y; r.c2 = 0.0f;
                    r.rden = 1.0f / r.c1;
                    r.flags = REC_LINEAR | ((p0y < p2y) ? REC_LIN_MINUS : 0u);   // :55
                    valid = true;
                }
            } else {
                r.a = a; r.b = p0y - p1y; r.c1 = p1y * p1y; r.c2 = p0y * p2y;    // :58, :60
                r.rden = 1.0f / a;
                r.flags = root ? REC_NEG_ROOT : 0u;
                valid = true;
            }
            if (valid) valid = accept_interval(r, p0y, p2y, lo, hi);
        }
        const unsigned long long m = __ballot(valid);
        if (valid) {
            const uint32_t pos = n_out + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            r.lo = lo; r.hi = hi;
            out_recs[out_base + pos] = r;
        }
        n_out += (uint32_t)__popcll(m);
    }
    if (lane == 0) glyph_rec_count[g] = n_out;
}

void launch_prepare(const int16_t *pts, const uint32_t *seg_p0, const uint32_t *glyph_seg_start,
                    uint32_t n_glyphs, Rec *out_recs, uint32_t *glyph_rec_count, hipStream_t stream)
{
    if (n_glyphs == 0) return;
    hipLaunchKernelGGL(prepare_kernel, dim3(n_glyphs), dim3(64), 0, stream, pts, seg_p0,
                       glyph_seg_start, n_glyphs, out_recs, glyph_rec_count);
}

}  // namespace fr
