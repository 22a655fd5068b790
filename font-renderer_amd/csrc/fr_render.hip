// fr_render.hip — the hot path: per-pixel winding / analytic coverage of quadratic
// Bezier contours on gfx950 (wave64, LDS-staged, no MFMA: not a dense contraction).
//
// Replaces the O(W*H*S) loop of renderGlyph + glyphWindingAt
// (/root/reference/src/tools/render_glyph.zig:24-31, :35-73) with a formulation
// that yields the SAME integers:
//
//   winding(cx, cy) = sum over accepted roots k of  sign_k * [ not (xx_k < cx) ]   (:54,:66)
//
// where (xx_k, sign_k) depend only on the segment and the ROW (cy).  So, per
// workgroup = (cell, band of 256/N pixel rows, column strip):
//   phase 0  stage the glyph's root records whose exact acceptance interval
//            [lo, hi] (fr_prepare.hip) meets the band into LDS — wave64 ballot +
//            prefix-popcount compaction;
//   phase 1  one lane per sample ROW: for each staged record with lo <= cy <= hi
//            evaluate t, xx, sign in the reference's own f32 operation order, turn
//            xx into J = #{sample columns j : cx(j) <= xx} against an LDS table of the
//            exact cx(j) (cx is monotone in j), sort the row's few crossings by J and
//            suffix-sum the signs -> breakpoints (b_i, winding on [b_{i-1}, b_i));
//   phase 2  one lane per 16-pixel window: spans -> sample bitmasks -> popcounts ->
//            16 output bytes in one coalesced 16-B store per lane (128 B per row run).
// Per-pixel work is O(crossings of its row), not O(segments).  Rows with more than
// `kmax` crossings fall back to the direct sum over records (same integers).
#include "fr_device.hpp"

namespace fr {

enum { MODE_WINDING_I16 = 0, MODE_GRAY_DEBUG = 1, MODE_MASK_NONZERO = 2, MODE_COVERAGE_U8 = 3 };

// direct sum over the glyph's records — the fallback for over-full rows
__device__ __noinline__ int brute_winding(const RecBounds *__restrict__ bounds,
                                          const RecPayload *__restrict__ payload, uint32_t n,
                                          float cx, float cy)
{
    int w = 0;
    for (uint32_t c = 0; c < n; ++c) {
        const RecBounds b = bounds[c];
        if (cy >= b.lo && cy <= b.hi) {
            float xx; int sgn;
            rec_cross(payload[c], cy, xx, sgn);
            if (!(xx < cx)) w += sgn;
        }
    }
    return w;
}

__device__ __forceinline__ uint8_t gray_debug(int w)
{
    int v = w * 20 + 100;                       // render_glyph.zig:28
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

template <int MODE, int N>
__global__ __launch_bounds__(256) void render_kernel(const RenderArgs A)
{
    constexpr uint32_t BAND = 256u / N;         // pixel rows per workgroup
    constexpr int WCOLS = 16 * N;               // sample columns per 16-pixel window
    extern __shared__ __align__(16) unsigned char smem[];

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t bid = blockIdx.x;
    const uint32_t strip = bid % A.strips; bid /= A.strips;
    const uint32_t band = bid % A.bands;
    const uint32_t jidx = bid / A.bands;
    const Job job = A.jobs[jidx];
    const uint32_t y0 = band * BAND, x0s = strip * A.strip_w;
    if (y0 >= job.h || x0s >= job.w) return;                    // workgroup-uniform
    const uint32_t band_rows = min(BAND, job.h - y0);
    const uint32_t sw = min(A.strip_w, job.w - x0s);            // strip width, pixels
    const uint32_t ncol = sw * N;                               // sample columns in the strip
    const uint32_t nrows = band_rows * N;                       // sample rows in the band
    const uint32_t kmax = A.kmax;
    const int phase = A.phase_center;

    float *s_cx = reinterpret_cast<float *>(smem);
    RecBounds *s_recb = reinterpret_cast<RecBounds *>(s_cx + (size_t)A.strip_w * N);
    RecPayload *s_recp = reinterpret_cast<RecPayload *>(s_recb + 256);
    uint32_t *s_list = reinterpret_cast<uint32_t *>(s_recp + 256);   // [kmax][256]
    uint32_t *s_meta = s_list + (size_t)kmax * 256u;                 // [256] first | cnt<<8 | ovf<<16
    uint32_t *s_nact = s_meta + 256;

    // exact sample abscissae of this strip: cx(j) = (f32(min_x + x) + off(i)) / scale   (:26)
    const int32_t min_xs = job.min_x + (int32_t)x0s;
    for (uint32_t j = tid; j < ncol; j += 256u)
        s_cx[j] = ((float)(min_xs + (int32_t)(j / N)) + sub_off((int)(j % N), N, phase)) / job.scale;

    // ray height of sample row r of the band: cy = (f32(max_y - y) - off(jj)) / scale  (:27)
    auto row_cy = [&](uint32_t r) -> float {
        const int32_t y = (int32_t)(y0 + r / N);
        return ((float)(job.max_y - y) - sub_off((int)(r % N), N, phase)) / job.scale;
    };
    const bool row_valid = tid < nrows;
    const float cy = row_cy(row_valid ? tid : 0u);
    const float cy_top = row_cy(0u), cy_bot = row_cy(nrows - 1u);  // cy is non-increasing in r

    const uint32_t g = job.glyph;
    const size_t rec_off = 2u * (size_t)A.glyph_seg_start[g];
    const uint32_t rec_cnt = A.glyph_rec_count[g];
    const RecBounds *gb = A.bounds + rec_off;
    const RecPayload *gp = A.payload + rec_off;

    // guess for J from the affine map, fixed up against the exact table
    const float jscale = job.scale * (float)N;
    const float joff = (float)min_xs * (float)N + (phase ? 0.5f : 0.0f) - 1.0f;

    uint32_t cnt = 0;
    for (uint32_t base = 0; base < rec_cnt; base += 256u) {
        if (tid == 0) *s_nact = 0;
        __syncthreads();
        // ---- phase 0: band-level cull + compaction into LDS
        const uint32_t c = base + tid;
        bool act = false;
        RecBounds b;
        if (c < rec_cnt) {
            b = gb[c];
            act = (b.hi >= cy_bot) && (b.lo <= cy_top);
        }
        const unsigned long long m = __ballot(act);
        uint32_t wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(s_nact, (uint32_t)__popcll(m));
        wbase = __shfl(wbase, 0);
        if (act) {
            const uint32_t idx = wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            s_recb[idx] = b;
            s_recp[idx] = gp[c];
        }
        __syncthreads();
        const uint32_t nact = *s_nact;
        // ---- phase 1: crossings of my sample row
        if (row_valid) {
            for (uint32_t k = 0; k < nact; ++k) {
                const RecBounds rb = s_recb[k];
                if (cy >= rb.lo && cy <= rb.hi) {
                    float xx; int sgn;
                    rec_cross(s_recp[k], cy, xx, sgn);
                    // J = #{ j in [0, ncol) : cx(j) <= xx }  ==  #{ j : not (xx < cx(j)) }   (:54,:66)
                    float gf = xx * jscale - joff;
                    gf = fminf(fmaxf(gf, 0.0f), (float)ncol);
                    int J = (int)gf;
                    while (J < (int)ncol && s_cx[J] <= xx) ++J;
                    while (J > 0 && s_cx[J - 1] > xx) --J;
                    if (J > 0) {
                        if (cnt < kmax) s_list[cnt * 256u + tid] = ((uint32_t)J << 16) | ((uint32_t)sgn & 0xffffu);
                        ++cnt;
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- sort my row's crossings by J, suffix-sum the signs into breakpoints
    {
        const bool ovf = cnt > kmax;
        uint32_t first = 0;
        if (!ovf && cnt > 0) {
            for (uint32_t i = 1; i < cnt; ++i) {
                const uint32_t e = s_list[i * 256u + tid];
                int j = (int)i - 1;
                while (j >= 0 && (s_list[(uint32_t)j * 256u + tid] >> 16) > (e >> 16)) {
                    s_list[(uint32_t)(j + 1) * 256u + tid] = s_list[(uint32_t)j * 256u + tid];
                    --j;
                }
                s_list[(uint32_t)(j + 1) * 256u + tid] = e;
            }
            uint32_t mpos = cnt;
            int run = 0;
            for (int i = (int)cnt - 1; i >= 0; --i) {
                const uint32_t e = s_list[(uint32_t)i * 256u + tid];
                run += (int)(int16_t)(e & 0xffffu);
                const uint32_t Jb = e >> 16;
                if (i == 0 || (s_list[(uint32_t)(i - 1) * 256u + tid] >> 16) != Jb) {
                    --mpos;
                    s_list[mpos * 256u + tid] = (Jb << 16) | ((uint32_t)run & 0xffffu);
                }
            }
            first = mpos;
        }
        s_meta[tid] = first | ((ovf ? 0u : cnt) << 8) | ((ovf ? 1u : 0u) << 16);
    }
    __syncthreads();

    // ---- phase 2: 16-pixel windows
    const uint32_t nwin = (sw + 15u) / 16u;
    const size_t esz = (MODE == MODE_WINDING_I16) ? 2 : 1;
    for (uint32_t win = tid; win < band_rows * nwin; win += 256u) {
        const uint32_t yl = win / nwin, wx = win % nwin;
        const uint32_t px0 = wx * 16u;                          // first pixel of the window in the strip
        const int j0 = (int)(px0 * N);
        const uint32_t nvalid = min(16u, sw - px0);
        const size_t eidx = ((size_t)job.out_y + y0 + yl) * A.out_stride + job.out_x + x0s + px0;

        if (MODE == MODE_COVERAGE_U8) {
            uint32_t k[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) k[p] = 0;
#pragma unroll
            for (int rr = 0; rr < N; ++rr) {
                const uint32_t r = yl * N + rr;
                const uint32_t meta = s_meta[r];
                unsigned long long mask = 0;
                if (meta >> 16) {
                    const float cyr = row_cy(r);
                    for (int cidx = 0; cidx < WCOLS; ++cidx) {
                        const int j = j0 + cidx;
                        if (j < (int)ncol && brute_winding(gb, gp, rec_cnt, s_cx[j], cyr) != 0)
                            mask |= 1ull << cidx;
                    }
                } else {
                    const uint32_t fst = meta & 0xffu, ecnt = (meta >> 8) & 0xffu;
                    int prev = 0;
                    for (uint32_t i = fst; i < ecnt; ++i) {
                        const uint32_t e = s_list[i * 256u + r];
                        const int bq = (int)(e >> 16);
                        if ((e & 0xffffu) != 0u) {
                            const int lo = max(prev - j0, 0), hi = min(bq - j0, WCOLS);
                            if (hi > lo) mask |= ((~0ull) >> (64 - (hi - lo))) << lo;
                        }
                        prev = bq;
                    }
                }
#pragma unroll
                for (int p = 0; p < 16; ++p)
                    k[p] += (uint32_t)__popc((uint32_t)(mask >> (p * N)) & ((1u << N) - 1u));
            }
            uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const uint32_t v = (2u * 255u * k[p] + (uint32_t)(N * N)) / (2u * (uint32_t)(N * N));
                pk[p >> 2] |= v << (8 * (p & 3));
            }
            uint8_t *dst = reinterpret_cast<uint8_t *>(A.out) + eidx;
            if (nvalid == 16u && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
                *reinterpret_cast<uint4 *>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            } else {
                for (uint32_t p = 0; p < nvalid; ++p) dst[p] = (uint8_t)(pk[p >> 2] >> (8 * (p & 3)));
            }
        } else {
            // N == 1: one sample per pixel, winding value needed
            int w[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) w[p] = 0;
            const uint32_t r = yl;
            const uint32_t meta = s_meta[r];
            if (meta >> 16) {
                const float cyr = row_cy(r);
                for (uint32_t p = 0; p < nvalid; ++p)
                    w[p] = brute_winding(gb, gp, rec_cnt, s_cx[j0 + (int)p], cyr);
            } else {
                const uint32_t fst = meta & 0xffu, ecnt = (meta >> 8) & 0xffu;
                int prev = 0;
                for (uint32_t i = fst; i < ecnt; ++i) {
                    const uint32_t e = s_list[i * 256u + r];
                    const int bq = (int)(e >> 16);
                    const int v = (int)(int16_t)(e & 0xffffu);
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        const int j = j0 + p;
                        if (j >= prev && j < bq) w[p] = v;
                    }
                    prev = bq;
                }
            }
            if (MODE == MODE_WINDING_I16) {
                int16_t *dst = reinterpret_cast<int16_t *>(A.out) + eidx;
                if (nvalid == 16u && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
                    uint32_t pk[8];
#pragma unroll
                    for (int p = 0; p < 8; ++p)
                        pk[p] = ((uint32_t)w[2 * p] & 0xffffu) | ((uint32_t)w[2 * p + 1] << 16);
                    reinterpret_cast<uint4 *>(dst)[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    reinterpret_cast<uint4 *>(dst)[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
                } else {
                    for (uint32_t p = 0; p < nvalid; ++p) dst[p] = (int16_t)w[p];
                }
            } else {
                uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const uint32_t v = (MODE == MODE_GRAY_DEBUG) ? gray_debug(w[p]) : (w[p] != 0 ? 255u : 0u);
                    pk[p >> 2] |= v << (8 * (p & 3));
                }
                uint8_t *dst = reinterpret_cast<uint8_t *>(A.out) + eidx;
                if (nvalid == 16u && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
                    *reinterpret_cast<uint4 *>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                } else {
                    for (uint32_t p = 0; p < nvalid; ++p) dst[p] = (uint8_t)(pk[p >> 2] >> (8 * (p & 3)));
                }
            }
        }
    }
    (void)esz;
}

size_t render_lds_bytes(uint32_t strip_w, int n, uint32_t kmax)
{
    return (size_t)strip_w * n * 4 + 256 * sizeof(RecBounds) + 256 * sizeof(RecPayload) +
           (size_t)kmax * 256 * 4 + 256 * 4 + 16;
}

template <int MODE, int N>
static hipError_t launch_one(const RenderArgs &a, dim3 grid, size_t lds, hipStream_t stream)
{
    auto kern = render_kernel<MODE, N>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_render(const RenderArgs &a, int mode, int n, hipStream_t stream)
{
    const size_t lds = render_lds_bytes(a.strip_w, n, a.kmax);
    const dim3 grid((uint32_t)((size_t)a.n_jobs * a.bands * a.strips));
    if (mode == MODE_COVERAGE_U8) {
        if (n == 1) return launch_one<MODE_COVERAGE_U8, 1>(a, grid, lds, stream);
        if (n == 2) return launch_one<MODE_COVERAGE_U8, 2>(a, grid, lds, stream);
        if (n == 4) return launch_one<MODE_COVERAGE_U8, 4>(a, grid, lds, stream);
        return hipErrorInvalidValue;
    }
    if (n != 1) return hipErrorInvalidValue;
    if (mode == MODE_WINDING_I16) return launch_one<MODE_WINDING_I16, 1>(a, grid, lds, stream);
    if (mode == MODE_GRAY_DEBUG) return launch_one<MODE_GRAY_DEBUG, 1>(a, grid, lds, stream);
    if (mode == MODE_MASK_NONZERO) return launch_one<MODE_MASK_NONZERO, 1>(a, grid, lds, stream);
    return hipErrorInvalidValue;
}

}  // namespace fr
