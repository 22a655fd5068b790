// fr_exact.hip — the reference's exact-integer winding path on gfx950:
// GlyphInfo.init / CurveType.init (/root/reference/src/tools/render_glyph.zig:96-146),
// windingInGlyph (:160-247), solve2RootsWinding (:249-277), solve1RootCrossing (:279-300).
//
// Bug-compatible by design (the reference marks it "still has some problem", :157-159):
// the switch below follows the Zig arm by arm, including the -1 of .down_inv_u (:226).
// Predicates are evaluated in 128-bit integers; the reference uses i64, which traps /
// is undefined once dy*abxy*abxy leaves i64 — identical wherever the reference is defined.
//
// Shape: one lane per query point; curves are staged through LDS 256 at a time and
// read back as wave-uniform broadcasts.  Integer-only, no divides, no MFMA.
#include "fr_device.hpp"

namespace fr {

enum : uint8_t {
    CT_X_AXIS = 0, CT_BALANCE, CT_UP_STRIGHT, CT_UP_NORMAL, CT_UP_U, CT_UP_INV_U,
    CT_DOWN_STRIGHT, CT_DOWN_NORMAL, CT_DOWN_INV_U, CT_DOWN_U
};

typedef __int128 i128;

__device__ __forceinline__ int iabs_(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int isign_(int v) { return (v > 0) - (v < 0); }

__device__ inline uint8_t curve_type(int p0x, int p0y, int p1x, int p1y, int p2x, int p2y)
{
    if (p0y == p2y) return (p1y == p0y) ? CT_X_AXIS : CT_BALANCE;                                  // :97
    if (p0y < p2y) {
        if (iabs_(p0x + p2x - 2 * p1x) <= 1 && iabs_(p0y + p2y - 2 * p1y) <= 1) return CT_UP_STRIGHT;   // :99
        if (p0y <= p1y && p1y <= p2y) return CT_UP_NORMAL;                                         // :100
        return (p1y < p0y) ? CT_UP_U : CT_UP_INV_U;                                                // :101
    }
    if (iabs_(p0x + p2x - 2 * p1x) <= 1 && iabs_(p0y + p2y - 2 * p1y) <= 1) return CT_DOWN_STRIGHT;     // :103
    if (p2y <= p1y && p1y <= p0y) return CT_DOWN_NORMAL;                                           // :104
    return (p1y > p0y) ? CT_DOWN_INV_U : CT_DOWN_U;                                                // :105
}

// one lane per curve.  seg_p0[s] = point index of the curve's p0, seg_prev[s] = point
// index of the previous curve's p0 in the same contour (wraps: :126-127).
__global__ __launch_bounds__(256) void glyph_info_kernel(const int16_t *__restrict__ pts,
                                                         const uint32_t *__restrict__ seg_p0,
                                                         const uint32_t *__restrict__ seg_prev,
                                                         uint32_t n_seg, int K, uint8_t *__restrict__ ctype,
                                                         uint8_t *__restrict__ inc_p0)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= n_seg) return;
    const int16_t *p = pts + 2u * (size_t)seg_p0[s];
    const int16_t *q = pts + 2u * (size_t)seg_prev[s];
    // K > 1: the same rules applied to the glyph scaled by K
    const int p_2y = K * q[1], p_1y = K * q[3];
    const int p0x = K * p[0], p0y = K * p[1], p1x = K * p[2], p1y = K * p[3], p2x = K * p[4], p2y = K * p[5];
    const int prev_end = 2 * isign_(p0y - p_1y) + isign_(p0y - p_2y);          // :132
    const int curr_start = 2 * isign_(p1y - p0y) + isign_(p2y - p0y);          // :133
    inc_p0[s] = (uint8_t)(curr_start != 0 && (prev_end == 0 || ((prev_end > 0) != (curr_start < 0))));   // :136
    ctype[s] = curve_type(p0x, p0y, p1x, p1y, p2x, p2y);                       // :137
}

struct Pt { int x, y; };

__device__ inline int solve2(Pt p, Pt p0, Pt p1, Pt p2)                         // :249-277
{
    const i128 ay = (i128)p0.y + p2.y - 2 * (i128)p1.y;
    const i128 by = ((i128)p1.y - p0.y) * 2;
    const i128 cy = (i128)p0.y - p.y;
    const i128 dy = by * by - 4 * ay * cy;
    if (dy <= 0) return 0;
    const i128 ax = (i128)p0.x + p2.x - 2 * (i128)p1.x;
    const i128 bx = ((i128)p1.x - p0.x) * 2;
    const i128 cx = (i128)p0.x - p.x;
    const i128 abxy = ax * by - ay * bx;
    if (abxy == 0) return 0;
    const i128 tmp = 2 * ay * (ax * cy - ay * cx) - by * abxy;
    const int s = abxy > 0 ? 1 : -1;
    if (tmp == 0) return s;
    const i128 l = dy * abxy * abxy, r = tmp * tmp;
    if (tmp > 0) return (l < r) ? 0 : s;
    return (l <= r) ? 0 : s;
}

__device__ inline bool solve1(Pt p, Pt p0, Pt p1, Pt p2, bool tilt_up)          // :279-300
{
    const i128 ay = (i128)p0.y + p2.y - 2 * (i128)p1.y;
    const i128 by = ((i128)p1.y - p0.y) * 2;
    const i128 cy = (i128)p0.y - p.y;
    const i128 dy = by * by - 4 * ay * cy;
    if (dy == 0) return false;                                                  // :285
    const i128 ax = (i128)p0.x + p2.x - 2 * (i128)p1.x;
    const i128 bx = ((i128)p1.x - p0.x) * 2;
    const i128 cx = (i128)p0.x - p.x;
    const i128 abxy = ax * by - ay * bx;
    const i128 tmp = 2 * ay * (ax * cy - ay * cx) - by * abxy;
    if (abxy == 0) return tmp <= 0;                                             // :292
    if ((abxy > 0) != tilt_up) {                                                // :293
        if (tmp <= 0) return true;
        return dy * abxy * abxy >= tmp * tmp;
    }
    if (tmp >= 0) return false;
    return dy * abxy * abxy <= tmp * tmp;
}

struct CurveLds {   // 32 B (coordinates of the K-times scaled glyph: up to 2^18)
    int32_t p0x, p0y, p1x, p1y, p2x, p2y;
    uint8_t type, inc;
    uint16_t pad;
    uint32_t pad2;
};

__device__ inline int curve_winding(const CurveLds &c, Pt p)                    // :170-243
{
    const Pt p0{c.p0x, c.p0y}, p1{c.p1x, c.p1y}, p2{c.p2x, c.p2y};
    const bool inc = c.inc != 0;
    int w = 0;
    switch (c.type) {
    case CT_X_AXIS: break;
    case CT_BALANCE:
        if (inc && p.y == p0.y) {
            if (p.x < p0.x) w += (p1.y < p0.y) ? 1 : -1;
        } else if ((!(p1.y < p0.y)) != (p.y < p0.y)) {
            w += solve2(p, p0, p1, p2);
        }
        break;
    case CT_UP_STRIGHT:
        if ((p0.y < p.y || (inc && p0.y == p.y)) && p.y < p2.y) {
            const long long v1 = (long long)(p.y - p0.y) * (p2.x - p0.x);
            const long long v2 = (long long)(p.x - p0.x) * (p2.y - p0.y);
            if (v1 >= v2) w += -1;
        }
        break;
    case CT_UP_NORMAL:
        if ((p0.y < p.y || (inc && p0.y == p.y)) && p.y < p2.y) {
            if (solve1(p, p0, p1, p2, true)) w += -1;
        }
        break;
    case CT_UP_U:
        if (p0.y <= p.y && p.y < p2.y) {
            const bool cross = solve1(p, p0, p1, p2, true);
            if (p0.y < p.y) {
                if (cross) w += -1;
            } else if (inc && (cross != (p.x <= p0.x))) {
                w += cross ? -1 : 1;
            }
        } else if (p.y < p0.y) {
            w += solve2(p, p0, p1, p2);
        }
        break;
    case CT_UP_INV_U:
        if ((p0.y < p.y || (inc && p0.y == p.y)) && p.y <= p2.y) {
            if (solve1(p, p0, p1, p2, true)) w += -1;
        } else if (p2.y < p.y) {
            w += solve2(p, p0, p1, p2);
        }
        break;
    case CT_DOWN_STRIGHT:
        if (p2.y < p.y && (p.y < p0.y || (inc && p.y == p0.y))) {
            const long long v1 = (long long)(p.y - p0.y) * (p2.x - p0.x);
            const long long v2 = (long long)(p.x - p0.x) * (p2.y - p0.y);
            if (v1 <= v2) w += 1;
        }
        break;
    case CT_DOWN_NORMAL:
        if (p2.y < p.y && (p.y < p0.y || (inc && p.y == p0.y))) {
            if (solve1(p, p0, p1, p2, false)) w += 1;
        }
        break;
    case CT_DOWN_INV_U:
        if (p2.y < p.y && p.y <= p0.y) {
            const bool cross = solve1(p, p0, p1, p2, false);
            if (p0.y > p.y) {
                if (cross) w += -1;                                             // :226 (as written)
            } else if (inc && (cross != (p.x > p0.x))) {
                w += cross ? 1 : -1;
            }
        } else if (p.y > p0.y) {
            w += solve2(p, p0, p1, p2);
        }
        break;
    case CT_DOWN_U:
        if (p2.y <= p.y && (p.y < p0.y || (inc && p.y == p0.y))) {
            if (solve1(p, p0, p1, p2, false)) w += 1;
        } else if (p.y < p2.y) {
            w += solve2(p, p0, p1, p2);
        }
        break;
    default: break;
    }
    return w;
}

// queries == nullptr: a lattice, query q -> (w, h) = (q % lat_w, q / lat_w), point
// (lat_x0 + w, lat_y0 - h); Image.GlyphDebug.render (Image.zig:227-236) is lat_x0 = x_min - 1,
// lat_y0 = y_max + 1.  The glyph's points are multiplied by K first (K = 1: the reference's own
// path; K = 2..8: the K-times refined lattice of SURVEY §8 f-3, whose integer points are the
// font-unit points (x / K, y / K)).
__global__ __launch_bounds__(256) void exact_winding_kernel(const int16_t *__restrict__ pts,
                                                            const uint32_t *__restrict__ seg_p0,
                                                            const uint8_t *__restrict__ ctype,
                                                            const uint8_t *__restrict__ inc_p0,
                                                            uint32_t n_seg,
                                                            const int16_t *__restrict__ queries,
                                                            uint64_t n_query, uint32_t lat_w,
                                                            int lat_x0, int lat_y0, int K,
                                                            int16_t *__restrict__ out)
{
    __shared__ CurveLds s_curve[256];
    const uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const bool valid = q < n_query;
    Pt p{0, 0};
    if (valid) {
        if (queries) { p.x = queries[2 * q]; p.y = queries[2 * q + 1]; }
        else { p.x = lat_x0 + (int)(q % lat_w); p.y = lat_y0 - (int)(q / lat_w); }
    }
    int w = 0;
    for (uint32_t base = 0; base < n_seg; base += 256u) {
        const uint32_t s = base + threadIdx.x;
        if (s < n_seg) {
            const int16_t *pp = pts + 2u * (size_t)seg_p0[s];
            CurveLds c;
            c.p0x = K * pp[0]; c.p0y = K * pp[1]; c.p1x = K * pp[2]; c.p1y = K * pp[3]; c.p2x = K * pp[4]; c.p2y = K * pp[5];
            c.type = ctype[s]; c.inc = inc_p0[s]; c.pad = 0; c.pad2 = 0;
            s_curve[threadIdx.x] = c;
        }
        __syncthreads();
        const uint32_t n = min(256u, n_seg - base);
        if (valid)
            for (uint32_t k = 0; k < n; ++k) w += curve_winding(s_curve[k], p);
        __syncthreads();
    }
    if (valid) out[q] = (int16_t)w;
}

void launch_glyph_info(const int16_t *pts, const uint32_t *seg_p0, const uint32_t *seg_prev,
                       uint32_t n_seg, int K, uint8_t *ctype, uint8_t *inc_p0, hipStream_t stream)
{
    if (!n_seg) return;
    hipLaunchKernelGGL(glyph_info_kernel, dim3((n_seg + 255u) / 256u), dim3(256), 0, stream, pts,
                       seg_p0, seg_prev, n_seg, K, ctype, inc_p0);
}

// n x n lattice points per pixel -> round_half_up(255 * inside / n^2), inside = winding != 0
// (the box-filtered non-zero fill of FR_COVERAGE_U8, on the exact-integer inside test)
__global__ __launch_bounds__(256) void exact_cover_kernel(const int16_t *__restrict__ wind, uint32_t w_px,
                                                          uint32_t h_px, uint32_t n, uint8_t *__restrict__ out)
{
    const uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (q >= (uint64_t)w_px * h_px) return;
    const uint32_t x = (uint32_t)(q % w_px), y = (uint32_t)(q / w_px);
    uint32_t inside = 0;
    for (uint32_t j = 0; j < n; ++j)
        for (uint32_t i = 0; i < n; ++i)
            inside += wind[((size_t)y * n + j) * ((size_t)w_px * n) + (size_t)x * n + i] != 0;
    out[q] = (uint8_t)((2u * 255u * inside + n * n) / (2u * n * n));
}

void launch_exact_cover(const int16_t *wind, uint32_t w_px, uint32_t h_px, uint32_t n, uint8_t *out, hipStream_t stream)
{
    const uint64_t nq = (uint64_t)w_px * h_px;
    if (!nq) return;
    hipLaunchKernelGGL(exact_cover_kernel, dim3((uint32_t)((nq + 255u) / 256u)), dim3(256), 0, stream, wind, w_px, h_px, n, out);
}

// Image.GlyphDebug.setWindingLinear (Image.zig:192-200): one RGB triple per lattice point —
//   c = |winding| *| scale (u16 saturating multiply); main = min(c, 255); sub = (c == main) ? 0 : overflow;
//   winding > 0: {sub, sub, main} (blue) else {main, sub, sub} (red; winding == 0 gives {0, 0, 0} either way)
__global__ __launch_bounds__(256) void glyph_debug_color_kernel(const int16_t *__restrict__ wind, uint64_t n,
                                                                uint32_t scale, uint32_t overflow, uint8_t *__restrict__ rgb)
{
    const uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (q >= n) return;
    const int v = wind[q];
    uint32_t c = (uint32_t)(v < 0 ? -v : v) * scale;
    if (c > 65535u) c = 65535u;
    const uint32_t mainc = c > 255u ? 255u : c, sub = (c == mainc) ? 0u : overflow;
    uint8_t *d = rgb + 3u * q;
    d[0] = (uint8_t)(v > 0 ? sub : mainc);
    d[1] = (uint8_t)sub;
    d[2] = (uint8_t)(v > 0 ? mainc : sub);
}

void launch_glyph_debug_color(const int16_t *wind, uint64_t n, uint32_t scale, uint32_t overflow, uint8_t *rgb, hipStream_t stream)
{
    if (!n) return;
    hipLaunchKernelGGL(glyph_debug_color_kernel, dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, stream, wind, n, scale, overflow, rgb);
}

void launch_exact_winding(const int16_t *pts, const uint32_t *seg_p0, const uint8_t *ctype,
                          const uint8_t *inc_p0, uint32_t n_seg, const int16_t *queries,
                          uint64_t n_query, uint32_t lat_w, int lat_x0, int lat_y0, int K, int16_t *out,
                          hipStream_t stream)
{
    if (!n_query) return;
    hipLaunchKernelGGL(exact_winding_kernel, dim3((uint32_t)((n_query + 255u) / 256u)), dim3(256), 0,
                       stream, pts, seg_p0, ctype, inc_p0, n_seg, queries, n_query, lat_w, lat_x0,
                       lat_y0, K, out);
}

}  // namespace fr
