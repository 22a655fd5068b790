// fr_records.hpp — root records with GUARANTEED acceptance brackets (shared by the stand-alone
// prepare_kernel and by the render kernel, which builds the records of its glyph in LDS).
//
// Why a bracket is sound.  For one root of one segment the reference decides "accept" from
// (render_glyph.zig:58-64)
//        delta = fl(fl(fl(cy*a) + c1) - c2);   reject if delta < 0
//        t     = fl(fl(B +/- sqrt(delta)) / a); reject if t < 0 or t >= 1
// with a, B, c1, c2 constants of the segment.  Every step is a correctly rounded IEEE
// operation with one varying operand, hence a monotone function of it; so delta(cy) is
// monotone (direction = sign a), sqrt is non-decreasing, and t+(cy) is non-decreasing /
// t-(cy) non-increasing for either sign of a.  The three rejection tests are therefore each a
// one-sided cut of the cy axis: classify() below is a monotone step function 0 -> 1 -> 2
// ("below the accepted set", "accepted", "above it") over the ordered binary32 values.
// Consequently ONE probe with class 0 proves that every smaller cy is rejected, and one probe
// with class 2 proves the same for every larger cy.  We probe the three heights where the
// cuts sit in exact arithmetic (t = 0 at p0y, t = 1 at p2y, delta = 0 at the vertex) and a few
// rounding-plateau widths either side of them, evaluating the reference's own expression, and
// keep  lo = just above the highest class-0 probe,  hi = just below the lowest class-2 probe.
// [lo, hi] CONTAINS the accepted set (exactly, not approximately); the render kernel culls
// with it and still applies the reference's own three tests to whatever survives, so the
// bracket only ever saves work.  The same holds for the a == 0 branch (:51-52).
#pragma once
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
// Only used when the cheap probes leave a side unbounded (rare): exponential, then bisection.
__device__ __noinline__ uint32_t first_at_least(float ra, float rb, float rc1, float rc2, uint32_t rflags, int T, uint32_t L, uint32_t H)
{
    // by value: passing the record by reference to a non-inlined function would force it into scratch
    Rec r;
    r.a = ra; r.b = rb; r.c1 = rc1; r.c2 = rc2; r.flags = rflags;
    uint32_t step = 1u;
    while (step < (1u << 30) && H - L > 2u * step) {
        const uint32_t pl = L + step - 1u;
        if (classify(r, key2f(pl)) >= T) { H = pl; break; }
        L = pl + 1u;
        const uint32_t ph = H - step;
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

__device__ inline bool accept_bracket(const Rec &r, float p0y, float p2y, float &lo, float &hi)
{
    const uint32_t kmin = f2key(-3.402823466e+38f), kmax = f2key(3.402823466e+38f);
    // LB: every key < LB is class 0.  HB: every key >= HB is class 2.
    uint32_t LB = kmin, HB = kmax + 1u;
    bool seen1 = false;
    auto probe = [&](float c) {
        if (!(c >= -3.402823466e+38f && c <= 3.402823466e+38f)) return;      // hints only
        const uint32_t k = f2key(c);
        const int cl = classify(r, c);
        if (cl == 0) { if (k + 1u > LB) LB = k + 1u; }
        else if (cl == 2) { if (k < HB) HB = k; }
        else seen1 = true;
    };
    // width of one rounding plateau of cy*a + c1 near the glyph: ulp(max|c1|,|c2|) / |a|
    float w = 0.0f;
    float vert = 0.0f;
    const bool quad = !(r.flags & REC_LINEAR);
    if (quad) {
        vert = p0y - (r.b * r.b) / r.a;                                       // vertex: y(B/a)
        const float big = fmaxf(fmaxf(fabsf(r.c1), fabsf(r.c2)), 1.0f);
        w = (big * 1.1920929e-07f) / fabsf(r.a);                              // ~ulp(big)/|a|
    }
    auto probe3 = [&](float c) {
        const float d = fmaxf(4.0f * w, fabsf(c) * 4.76837158e-07f);          // >= 4 plateaus, >= 4 ulp
        probe(c - d);
        probe(c);
        probe(c + d);
    };
    probe3(p0y);                                     // (no array, no dynamic index: stays in registers)
    probe3(p2y);
    if (quad) probe3(vert);
    if (LB >= HB) return false;                          // class 0 up to LB-1, class 2 from HB: never accepted
    // a side the probes did not bound (no class-0 / class-2 probe at all): settle it exactly
    if (LB == kmin) LB = first_at_least(r.a, r.b, r.c1, r.c2, r.flags, 1, kmin, HB);
    if (HB == kmax + 1u) HB = first_at_least(r.a, r.b, r.c1, r.c2, r.flags, 2, LB, kmax + 1u);
    if (LB >= HB) return false;
    (void)seen1;
    lo = key2f(LB);
    hi = key2f(HB - 1u);
    return true;
}


// Candidate root `root` (0: t+, or the single root of an a == 0 segment; 1: t-) of the segment
// whose three control points are p[0..5] (i16 x,y pairs).  Returns false when the root does not
// exist or its bracket is provably empty.
__device__ inline bool build_record(const int16_t *p, uint32_t root, Rec &r)
{
    // i16 -> f32 (render_glyph.zig:43-45)
    const float p0x = (float)p[0], p0y = (float)p[1];
    const float p1x = (float)p[2], p1y = (float)p[3];
    const float p2x = (float)p[4], p2y = (float)p[5];
    const float a = p0y - 2 * p1y + p2y;                 // :48
    r.ax = p0x - 2 * p1x + p2x;                          // :53/:65
    r.bx = 2 * (p1x - p0x);
    r.p0x = p0x;
    r.sqsign = root ? 0x80000000u : 0u;
    if (a == 0.0f) {                                     // :49
        if (root != 0u || p2y == p0y) return false;      // :50
        r.a = 0.0f; r.b = p0y; r.c1 = p2y - p0y; r.c2 = 0.0f;
        r.rden = 1.0f / r.c1;
        r.flags = REC_LINEAR | ((p0y < p2y) ? REC_LIN_MINUS : REC_LIN_PLUS);   // :55
    } else {
        r.a = a; r.b = p0y - p1y; r.c1 = p1y * p1y; r.c2 = p0y * p2y;    // :58, :60
        r.rden = 1.0f / a;
        r.flags = root ? REC_NEG_ROOT : 0u;
    }
    float lo, hi;
    if (!accept_bracket(r, p0y, p2y, lo, hi)) return false;
    r.lo = lo; r.hi = hi;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Row form of the same bracket, used by the render kernel's in-LDS record build.  A cell's sample
// rows have ray heights cy(r) = (f32(max_y - y) - off) / scale (render_glyph.zig:27), non-increasing
// in r, so by the monotonicity argued above the class of row r is NON-INCREASING in r (2 = above
// the accepted set, 1 = accepted, 0 = below) and the accepted rows are ONE half-open range
//        [ra, re),   ra = min{ r : class(r) <= 1 },   re = min{ r >= ra : class(r) == 0 }.
// Both ends are settled by evaluating the reference's own expression at the rows either side of a
// guess and walking while the class says so — the guess (where t = 0, t = 1 and delta = 0 sit in
// exact arithmetic) only decides how many rows are looked at, never the result.  Two candidates
// are discarded without a probe, by a bound that holds for the rounded values too:
//   * the root on the far side of a vertex at t_v = B/a >= 1 is >= fl(B/a) >= 1  (always rejected, :64)
//   * the root on the near side of a vertex at t_v < 0 is <= fl(B/a) < 0          (always rejected, :64)
// (B +/- sqrt(delta) is >= / <= B after rounding because B is representable, and a correctly
// rounded division by a is monotone; |B/a| >= 2^-17 cannot round to zero.)
struct RowGeom {
    int32_t max_y;
    float scale;
    uint32_t rows;       // sample rows of the cell (h * N)
    int n, phase;
    const float *cyt = nullptr;   // optional: the ray heights of all the cell's rows, computed once by this very expression
    __device__ __forceinline__ float cy(uint32_t r) const
    {
        if (cyt) return cyt[r];
        return ((float)(max_y - (int32_t)(r / (uint32_t)n)) - sub_off((int)(r % (uint32_t)n), n, phase)) / scale;
    }
    // fractional row index at which the ray height equals c (rows r >= x lie at or below c)
    __device__ __forceinline__ float row_of(float c) const
    {
        return ((float)max_y - c * scale) * (float)n - (phase ? 0.5f : 0.0f);
    }
};

// branch-free class of ray height cy for either kind of record; t by div_by_int (== IEEE `/`, fr_device.hpp)
__device__ __forceinline__ int classify_row(const Rec &r, float cy)
{
    const bool lin = (r.flags & REC_LINEAR) != 0;
    const float delta = cy * r.a + r.c1 - r.c2;                                  // :58
    const float sq = sqrt_rn(delta);
    const float num = lin ? (cy - r.b) : ((r.flags & REC_NEG_ROOT) ? (r.b - sq) : (r.b + sq));   // :51 / :60-61
    const float t = div_by_int(num, lin ? r.c1 : r.a, r.rden);
    // t falls with cy for the t- root and for a descending line
    const bool falls = lin ? !(r.c1 > 0.0f) : ((r.flags & REC_NEG_ROOT) != 0);
    int cl = 1;
    if (t >= 1.0f) cl = falls ? 0 : 2;                                           // :52 / :64
    if (t < 0.0f) cl = falls ? 2 : 0;
    if (!lin && delta < 0.0f) cl = (r.a > 0.0f) ? 0 : 2;                         // :59
    return cl;
}

// As build_record, but the bracket is this cell's exact sample-row range, written as integers
// into the lo / hi slots of the record (ra, re).  An empty range is ra = 1, re = 0.
// In three pieces, so that a set-up with lanes to spare can look at the four rows around the guessed
// ends in parallel (fr_c4.hpp): record_prep (fields + guesses), record_settle (the exact walk from the
// guesses), and their composition build_record_rows.
struct RowGuess {
    uint32_t ra, re;     // guessed first accepted row / first row past the range, clamped to [0, rows]
    bool empty;          // the candidate is discarded without a probe (see above)
};

__device__ __forceinline__ void record_prep(const int16_t *p, uint32_t root, const RowGeom &G, Rec &r, RowGuess &g)
{
    const float p0x = (float)p[0], p0y = (float)p[1];
    const float p1x = (float)p[2], p1y = (float)p[3];
    const float p2x = (float)p[4], p2y = (float)p[5];
    const float a = p0y - 2 * p1y + p2y;                 // :48
    r.ax = p0x - 2 * p1x + p2x;                          // :53/:65
    r.bx = 2 * (p1x - p0x);
    r.p0x = p0x;
    r.sqsign = root ? 0x80000000u : 0u;
    const bool lin = a == 0.0f;                          // :49
    const float b = p0y - p1y;
    // ends of the accepted set in exact arithmetic: `start` is where t = 0 (or the vertex), `stop`
    // where t = 1 (or the vertex); the vertex and t = 0 are accepted, t = 1 is not
    bool empty;
    float c_start = p0y, c_stop = p2y;
    bool stop_incl = false;
    if (lin) {
        empty = (root != 0u) || (p2y == p0y);            // :50
        r.a = 0.0f; r.b = p0y; r.c1 = p2y - p0y; r.c2 = 0.0f;
        r.rden = 1.0f / r.c1;
        r.flags = REC_LINEAR | ((p0y < p2y) ? REC_LIN_MINUS : REC_LIN_PLUS);   // :55
    } else {
        r.a = a; r.b = b; r.c1 = p1y * p1y; r.c2 = p0y * p2y;        // :58, :60
        r.rden = 1.0f / a;
        r.flags = root ? REC_NEG_ROOT : 0u;
        const bool far_side = (root == 0u) == (a > 0.0f);            // this root satisfies (t - t_v) * 1 >= 0
        const bool tv_le0 = (b * a) <= 0.0f, tv_lt0 = (b * a) < 0.0f;     // a, b are integers: signs are exact
        const bool tv_ge1 = (a > 0.0f) ? (b >= a) : (b <= a);
        const float yv = p0y - (b * b) * r.rden;         // vertex height (a guess only: no second division)
        if (far_side) {
            empty = tv_ge1;
            if (!tv_le0) c_start = yv;
        } else {
            empty = tv_lt0;
            if (!tv_ge1) { c_stop = yv; stop_incl = true; }
        }
    }
    g.empty = empty;
    g.ra = 1u; g.re = 0u;
    if (!empty) {
        // guesses: first row at / strictly below an end
        const float top = (float)G.rows;
        const bool start_hi = c_start >= c_stop;
        const float xh = G.row_of(start_hi ? c_start : c_stop), xl = G.row_of(start_hi ? c_stop : c_start);
        const bool hi_incl = start_hi ? true : stop_incl, lo_incl = start_hi ? stop_incl : true;
        const float fa = hi_incl ? __builtin_ceilf(xh) : __builtin_floorf(xh) + 1.0f;    // first row accepted
        const float fe = lo_incl ? __builtin_floorf(xl) + 1.0f : __builtin_ceilf(xl);    // first row past it
        g.ra = (uint32_t)fminf(fmaxf(fa, 0.0f), top);
        g.re = (uint32_t)fminf(fmaxf(fe, 0.0f), top);
    }
}

// the exact ends from any starting guess: walk while the class of the row says so
__device__ __forceinline__ void record_settle(const Rec &r, const RowGeom &G, uint32_t &ra, uint32_t &re)
{
    // (one trip each when the guess is right: keep the compiler from unrolling them)
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    while (ra > 0u && classify_row(r, G.cy(ra - 1u)) <= 1) --ra;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    while (ra < G.rows && classify_row(r, G.cy(ra)) == 2) ++ra;
    if (re < ra) re = ra;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    while (re > ra && classify_row(r, G.cy(re - 1u)) == 0) --re;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    while (re < G.rows && classify_row(r, G.cy(re)) >= 1) ++re;
}

__device__ __forceinline__ void build_record_rows(const int16_t *p, uint32_t root, const RowGeom &G, Rec &r)
{
    RowGuess g;
    record_prep(p, root, G, r, g);
    uint32_t ra = g.ra, re = g.re;
    if (!g.empty) record_settle(r, G, ra, re);
    r.lo = __builtin_bit_cast(float, ra);
    r.hi = __builtin_bit_cast(float, re);
}

}  // namespace fr
