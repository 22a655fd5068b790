// fr_render.hip — the hot path: per-pixel winding / analytic coverage of quadratic
// Bezier contours on gfx950 (wave64, LDS-staged, no MFMA: not a dense contraction).
//
// Replaces the O(W*H*S) loop of renderGlyph + glyphWindingAt
// (/root/reference/src/tools/render_glyph.zig:24-31, :35-73) with a formulation
// that yields the SAME integers:
//
//   winding(cx, cy) = sum over accepted roots k of  sign_k * [ not (xx_k < cx) ]   (:54,:66)
//
// where (xx_k, sign_k) depend only on the segment and the ROW (cy).  Per workgroup =
// (cell, band of 256/N pixel rows, column strip), 4 waves:
//   phase 0  stage the glyph's root records whose exact acceptance interval [lo, hi]
//            (fr_prepare.hip) meets the band into LDS — wave64 ballot + prefix-popcount
//            compaction;
//   phase 1  one lane per sample ROW.  Each wave pulls 64 staged records into registers
//            (one per lane) and walks them with v_readlane broadcasts — no LDS round trip
//            per record.  A lane whose cy lies in [lo, hi] evaluates t, xx, sign in the
//            reference's own f32 operation order, converts xx into
//            J = #{sample columns j : cx(j) <= xx} against an LDS table of the exact cx(j)
//            (cx is monotone in j), and inserts (J, sign) into a small SORTED array kept
//            in registers (min/max compare-exchange chain);
//   phase 1b the lane suffix-sums its signs from the right: crossing i TOGGLES
//            inside/outside iff the running winding changes between zero and non-zero
//            across it.  Each toggle at column t XORs a prefix mask into the 64-bit LDS
//            word of the 16-pixel window holding t (ds_xor_b64) and flips a per-row
//            "windows to the left are filled" parity word — O(toggles) per row;
//   phase 2  one lane per 16-pixel window: N mask words -> SWAR popcount per pixel ->
//            16 output bytes, one coalesced 16-B store per lane (256 B per row run).
// Per-pixel work is O(crossings of its row), not O(segments).  A row with more than CAP
// crossings falls back to the direct sum over records (same integers, slower).
#include "fr_device.hpp"

namespace fr {

enum { MODE_WINDING_I16 = 0, MODE_GRAY_DEBUG = 1, MODE_MASK_NONZERO = 2, MODE_COVERAGE_U8 = 3 };

// direct sum over the glyph's records — the fallback for over-full rows
__device__ __noinline__ int brute_winding(const Rec *__restrict__ recs, uint32_t n, float cx, float cy)
{
    int w = 0;
    for (uint32_t c = 0; c < n; ++c) {
        const Rec r = recs[c];
        if (cy >= r.lo && cy <= r.hi) {
            float xx; int sgn;
            rec_cross(r, cy, xx, sgn);
            if (!(xx < cx)) w += sgn;
        }
    }
    return w;
}

__device__ __forceinline__ uint32_t gray_debug(int w)
{
    int v = w * 20 + 100;                       // render_glyph.zig:28
    return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__device__ __forceinline__ float bcast(float v, uint32_t k)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)k));
}

constexpr uint32_t EMPTY = 0xffffffffu;

// Diagnostic build only (make STAMPS=1 -> libfr_raster_stamps.so): per-phase shader-clock
// sums of wave 0 of every workgroup, added to a buffer nothing else reads.  The shipped
// library is built without FR_STAMPS and executes no stamp.
#ifdef FR_STAMPS
__device__ unsigned long long g_stamps[8];
#define STAMP(i)                                                                          \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        unsigned long long t_;                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if (tid == 0) atomicAdd(&g_stamps[i], t_ - t_prev_);                              \
        t_prev_ = t_;                                                                     \
    } while (0)
#define STAMP_INIT()                                                                      \
    unsigned long long t_prev_;                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev_)::"memory")
#else
#define STAMP(i) do {} while (0)
#define STAMP_INIT() do {} while (0)
#endif

// LDS line of sample row r in the window-mask array.  Swapping line parity with bit 2
// puts rows r and r+4 (the same sub-row of two adjacent pixel rows, read together by one
// ds_read_b64 in phase 2) into different 128-B halves of the 256-B bank space.
__device__ __forceinline__ uint32_t mask_line(uint32_t r) { return r ^ ((r >> 2) & 1u); }

template <int MODE, int N, int CAP>
__global__ __launch_bounds__(256) void render_kernel(const RenderArgs A)
{
    constexpr uint32_t BAND = 256u / N;         // pixel rows per band
    constexpr int WCOLS = 16 * N;               // sample columns per 16-pixel window (<= 64)
    constexpr int WSHIFT = (N == 4) ? 6 : (N == 2 ? 5 : 4);
    constexpr unsigned long long WALL = (N == 4) ? ~0ull : ((1ull << WCOLS) - 1ull);
    constexpr bool COV = (MODE == MODE_COVERAGE_U8);
    extern __shared__ __align__(16) unsigned char smem[];

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    STAMP_INIT();
    uint32_t bid = blockIdx.x;
    const uint32_t strip = bid % A.strips; bid /= A.strips;
    const uint32_t bgrp = bid % A.band_groups;
    const uint32_t jidx = bid / A.band_groups;
    const Job job = A.jobs[jidx];
    const uint32_t x0s = strip * A.strip_w;
    const uint32_t band_first = bgrp * A.bands_per_wg;
    if (band_first * BAND >= job.h || x0s >= job.w) return;     // workgroup-uniform
    const uint32_t band_end = min(band_first + A.bands_per_wg, (job.h + BAND - 1u) / BAND);
    const uint32_t sw = min(A.strip_w, job.w - x0s);            // strip width, pixels
    const uint32_t ncol = sw * N;                               // sample columns in the strip
    const int phase = A.phase_center;
    const uint32_t nwin_log = A.nwin_log;                       // windows per row, padded to 2^k
    const uint32_t nwin_pad = 1u << nwin_log;

    // this lane's record of the glyph (record tid of the current 256-record chunk): issued
    // first so its latency hides under the cx-table divisions; for glyphs of <= 256 records
    // (the common case) it is loaded ONCE and re-culled per band from registers
    const uint32_t g = job.glyph;
    const Rec *grec = A.recs + 2u * (size_t)A.glyph_seg_start[g];
    const uint32_t rec_cnt = A.glyph_rec_count[g];
    Rec rg;
    rg.lo = 1.0f; rg.hi = 0.0f;                 // empty interval
    if (tid < rec_cnt) rg = grec[tid];

    // LDS: cx table (lives for the whole cell) | region R | s_fill[256] | s_nact[2].
    // Region R holds the staged records during phase 1 and is reused by the window masks
    // (coverage) / breakpoint rows (winding modes) of phase 1b/2.
    // s_cxp[0] = -inf, s_cxp[1 + j] = cx(j), s_cxp[1 + ncol] = +inf
    float *s_cxp = reinterpret_cast<float *>(smem);
    unsigned char *regionR = smem + A.lds_region;
    Rec *s_rec = reinterpret_cast<Rec *>(regionR);
    unsigned long long *s_mask = reinterpret_cast<unsigned long long *>(regionR);   // [256][nwin_pad]
    uint32_t *s_row = reinterpret_cast<uint32_t *>(regionR);                        // [256][CAP]
    uint32_t *s_fill = reinterpret_cast<uint32_t *>(smem + A.lds_tail);             // [256]
    uint32_t *s_nact = s_fill + 256;                                                // [2], zeroed below
    float *s_cyr = reinterpret_cast<float *>(s_nact + 2);                           // [2] band's top / bottom cy

    // exact sample abscissae of this strip: cx(j) = (f32(min_x + x) + off(i)) / scale   (:26)
    const int32_t min_xs = job.min_x + (int32_t)x0s;
    auto col_cx = [&](uint32_t j) -> float {
        return ((float)(min_xs + (int32_t)(j / N)) + sub_off((int)(j % N), N, phase)) / job.scale;
    };
    for (uint32_t j = tid; j < ncol; j += 256u) s_cxp[1u + j] = col_cx(j);
    if (tid < 2) s_nact[tid] = 0;
    if (tid == 2) s_cxp[0] = -__builtin_inff();
    if (tid == 3) s_cxp[1u + ncol] = __builtin_inff();

    // guess for J from the affine map, fixed up against the exact table
    const float jscale = job.scale * (float)N;
    const float joff = (float)min_xs * (float)N + (phase ? 0.5f : 0.0f) - 1.0f;
    const size_t out_col0 = (size_t)job.out_x + x0s;
    uint32_t flip = 0;                          // which s_nact counter the next chunk uses
    STAMP(0);                                   // setup: job, record load, cx table

  for (uint32_t band = band_first; band < band_end; ++band) {
    const uint32_t y0 = band * BAND;
    const uint32_t band_rows = min(BAND, job.h - y0);
    const uint32_t nrows = band_rows * N;                       // sample rows in the band

    // ray height of sample row r of the band: cy = (f32(max_y - y) - off(jj)) / scale  (:27)
    auto row_cy = [&](uint32_t r) -> float {
        const int32_t y = (int32_t)(y0 + r / N);
        return ((float)(job.max_y - y) - sub_off((int)(r % N), N, phase)) / job.scale;
    };
    const bool row_valid = tid < nrows;
    const float cy = row_cy(row_valid ? tid : 0u);
    // cy is non-increasing in r: the band spans [cy(nrows-1), cy(0)], this wave [wcy_bot, wcy_top]
    if (tid == 0) s_cyr[0] = cy;
    if (tid == nrows - 1u) s_cyr[1] = cy;
    const uint32_t wrow0 = tid & ~63u;
    const bool wave_has_rows = wrow0 < nrows;
    const float wcy_top = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cy)));
    const float wcy_bot = bcast(cy, wave_has_rows ? min(63u, nrows - 1u - wrow0) : 0u);

    uint32_t e[CAP];                            // sorted ascending: (J << 16) | (sign & 0xffff)
#pragma unroll
    for (int i = 0; i < CAP; ++i) e[i] = EMPTY;
    bool ovf = false;

    for (uint32_t base = 0; base < rec_cnt; base += 256u) {
        if (rec_cnt > 256u && !(base == 0 && band == band_first)) {     // multi-chunk glyph: reload
            rg.lo = 1.0f; rg.hi = 0.0f;
            if (base + tid < rec_cnt) rg = grec[base + tid];
        }
        // region R is free (previous chunk's walk / previous band's phase 2 are done) and the
        // cx table, s_nact zeroes are visible
        __syncthreads();
        // ---- phase 0: band-level cull from registers + compaction into LDS
        const float cy_top = s_cyr[0], cy_bot = s_cyr[1];
        const bool act = (rg.hi >= cy_bot) && (rg.lo <= cy_top) && (rg.lo <= rg.hi);
        const unsigned long long m = __ballot(act);
        uint32_t wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&s_nact[flip], (uint32_t)__popcll(m));
        wbase = __shfl(wbase, 0);
        if (act) s_rec[wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = rg;
        if (tid == 0) s_nact[flip ^ 1u] = 0;    // everyone read it before the barrier above
        __syncthreads();
        STAMP(1);                               // phase 0: cull + compaction
        const uint32_t nact = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_nact[flip]);
        flip ^= 1u;
        // ---- phase 1: crossings of my sample row; records travel lane -> wave by readlane
        for (uint32_t cb = 0; cb < nact && wave_has_rows; cb += 64u) {
            const Rec mine = s_rec[min(cb + lane, nact - 1u)];
            // records of this 64-chunk whose interval meets THIS wave's rows: one ballot, then a
            // scalar find-first-set loop — records that miss the wave cost nothing
            unsigned long long todo = __ballot((cb + lane < nact) && (mine.hi >= wcy_bot) && (mine.lo <= wcy_top));
            while (todo) {
                const uint32_t k = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1ull;
                const float lo = bcast(mine.lo, k), hi = bcast(mine.hi, k);
                if (row_valid && cy >= lo && cy <= hi) {
                    Rec r;
                    r.a = bcast(mine.a, k); r.b = bcast(mine.b, k); r.c1 = bcast(mine.c1, k);
                    r.c2 = bcast(mine.c2, k); r.ax = bcast(mine.ax, k); r.bx = bcast(mine.bx, k);
                    r.p0x = bcast(mine.p0x, k);
                    r.flags = (uint32_t)__builtin_amdgcn_readlane((int)mine.flags, (int)k);
                    float xx; int sgn;
                    rec_cross(r, cy, xx, sgn);
                    // J = #{ j in [0, ncol) : cx(j) <= xx }  ==  #{ j : not (xx < cx(j)) }   (:54,:66)
                    // <=> s_cxp[J] <= xx < s_cxp[J + 1] in the padded table; guess from the affine
                    // map, confirm with one paired read, walk only if the guess is off
                    float gf = xx * jscale - joff;
                    gf = fminf(fmaxf(gf, 0.0f), (float)ncol);
                    int J = (int)gf;
                    {
                        const float c0 = s_cxp[J], c1 = s_cxp[J + 1];
                        if (!(c0 <= xx && xx < c1)) {
                            while (s_cxp[J + 1] <= xx) ++J;          // +inf sentinel stops it at ncol
                            while (s_cxp[J] > xx) --J;               // -inf sentinel stops it at 0
                        }
                    }
                    if (J > 0) {
                        uint32_t x = ((uint32_t)J << 16) | ((uint32_t)sgn & 0xffffu);
                        // sorted insert, registers only: compare-exchange chain, 8 slots at a
                        // time; stop (wave-uniform) once every lane's carry is the EMPTY sentinel
#pragma unroll
                        for (int ch = 0; ch < CAP / 8; ++ch) {
#pragma unroll
                            for (int i = ch * 8; i < ch * 8 + 8; ++i) {
                                const uint32_t mn = min(e[i], x);
                                x = max(e[i], x);
                                e[i] = mn;
                            }
                            if (__ballot(x != EMPTY) == 0ull) break;
                        }
                        if (x != EMPTY) ovf = true;
                    }
                }
            }
        }
        STAMP(2);                               // phase 1: record walk (this wave)
    }
    __syncthreads();                            // every wave is done with the staged records
    STAMP(3);                                   // phase 1: waiting for the slowest wave

    const size_t out_row0 = (size_t)job.out_y + y0;

    if (COV) {
        // ---- phase 1b: my row's toggles -> window masks (LDS, XOR), fill parity (register)
        {
            // zero the mask array cooperatively (consecutive 16-B stores: no bank conflicts)
            uint4 *z = reinterpret_cast<uint4 *>(regionR);
            for (uint32_t q = tid; q < (128u << nwin_log); q += 256u) z[q] = make_uint4(0, 0, 0, 0);
            __syncthreads();
            unsigned long long *line = s_mask + ((size_t)mask_line(tid) << nwin_log);
            uint32_t fill = 0;
            if (!ovf) {
                int run = 0;
#pragma unroll
                for (int ch = CAP / 8 - 1; ch >= 0; --ch) {
                    if (__ballot(e[ch * 8] != EMPTY) == 0ull) continue;     // sorted: chunk empty in every lane
#pragma unroll
                    for (int i = ch * 8 + 7; i >= ch * 8; --i) {
                        if (e[i] != EMPTY) {
                            const int before = run;
                            run += (int)(int16_t)(e[i] & 0xffffu);
                            if ((run != 0) != (before != 0)) {
                                const uint32_t t = e[i] >> 16;                  // 1 .. ncol
                                const uint32_t wv = (t - 1u) >> WSHIFT;
                                const uint32_t cb = t - (wv << WSHIFT);         // 1 .. WCOLS bits set
                                atomicXor(line + wv, ~0ull >> (64u - cb));
                                fill ^= (1u << wv) - 1u;                        // every window to the left flips
                            }
                        }
                    }
                }
            } else if (row_valid) {
                // over-full row: direct sum per sample column (same integers, slow path)
                for (uint32_t wv = 0; wv < nwin_pad; ++wv) {
                    unsigned long long mk = 0;
                    for (int cidx = 0; cidx < WCOLS; ++cidx) {
                        const uint32_t j = (wv << WSHIFT) + (uint32_t)cidx;
                        if (j < ncol && brute_winding(grec, rec_cnt, col_cx(j), cy) != 0) mk |= 1ull << cidx;
                    }
                    line[wv] = mk;
                }
            }
            s_fill[tid] = fill;
        }
        STAMP(4);                               // phase 1b: zero + toggles
        __syncthreads();
        STAMP(5);

        // ---- phase 2: one lane per 16-pixel window
        const uint32_t wx = tid & (nwin_pad - 1u);
        const uint32_t px0 = wx * 16u;
        if (px0 < sw) {
            const uint32_t nvalid = min(16u, sw - px0);
            for (uint32_t yl = tid >> nwin_log; yl < band_rows; yl += (256u >> nwin_log)) {
                unsigned long long mask[N];
#pragma unroll
                for (int rr = 0; rr < N; ++rr) {
                    const uint32_t r = yl * N + rr;
                    const unsigned long long mk = s_mask[((size_t)mask_line(r) << nwin_log) + wx];
                    mask[rr] = mk ^ (((s_fill[r] >> wx) & 1u) ? WALL : 0ull);
                }
                uint32_t pk[4];
                if (N == 4) {
                    // SWAR: per-nibble popcounts of the 4 sample rows, summed per pixel
                    unsigned long long s01 = 0, s23 = 0;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        unsigned long long x = mask[rr];
                        x = x - ((x >> 1) & 0x5555555555555555ull);
                        x = (x & 0x3333333333333333ull) + ((x >> 2) & 0x3333333333333333ull);
                        if (rr < 2) s01 += x; else s23 += x;
                    }
                    const unsigned long long M = 0x0f0f0f0f0f0f0f0full;
                    const unsigned long long ke = (s01 & M) + (s23 & M);                 // even pixels, 0..16
                    const unsigned long long ko = ((s01 >> 4) & M) + ((s23 >> 4) & M);   // odd pixels
                    // round_half_up(255*k/16) = 16k - (k > 8), per byte, in independent 32-bit halves
                    uint32_t ve[2], vo[2];
#pragma unroll
                    for (int hlf = 0; hlf < 2; ++hlf) {
                        const uint32_t a = (uint32_t)(ke >> (32 * hlf)), b2 = (uint32_t)(ko >> (32 * hlf));
                        ve[hlf] = (a << 4) - (((a + 0x07070707u) >> 4) & 0x01010101u);
                        vo[hlf] = (b2 << 4) - (((b2 + 0x07070707u) >> 4) & 0x01010101u);
                    }
                    pk[0] = __builtin_amdgcn_perm(vo[0], ve[0], 0x05010400u);
                    pk[1] = __builtin_amdgcn_perm(vo[0], ve[0], 0x07030602u);
                    pk[2] = __builtin_amdgcn_perm(vo[1], ve[1], 0x05010400u);
                    pk[3] = __builtin_amdgcn_perm(vo[1], ve[1], 0x07030602u);
                } else {
                    pk[0] = pk[1] = pk[2] = pk[3] = 0;
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        uint32_t k = 0;
#pragma unroll
                        for (int rr = 0; rr < N; ++rr)
                            k += (uint32_t)__popc((uint32_t)(mask[rr] >> (p * N)) & ((1u << N) - 1u));
                        const uint32_t v = (2u * 255u * k + (uint32_t)(N * N)) / (2u * (uint32_t)(N * N));
                        pk[p >> 2] |= v << (8 * (p & 3));
                    }
                }
                uint8_t *dst = reinterpret_cast<uint8_t *>(A.out) + (out_row0 + yl) * A.out_stride + out_col0 + px0;
                if (nvalid == 16u && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
                    *reinterpret_cast<uint4 *>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                } else {
                    for (uint32_t p = 0; p < nvalid; ++p) dst[p] = (uint8_t)(pk[p >> 2] >> (8 * (p & 3)));
                }
            }
        }
    } else {
        // ---- winding-value modes (N == 1): breakpoints (b_i, winding on [b_{i-1}, b_i)) per row
        {
            uint32_t *dst = s_row + tid * CAP;
            int run = 0;
#pragma unroll
            for (int i = CAP - 1; i >= 0; --i) {
                const bool have = e[i] != EMPTY;
                if (have) run += (int)(int16_t)(e[i] & 0xffffu);
                dst[i] = have ? ((e[i] & 0xffff0000u) | ((uint32_t)run & 0xffffu)) : 0xffff0000u;
            }
            if (ovf) dst[0] = 0xffffffffu;                      // (b = 0xffff, v = -1) cannot occur otherwise
        }
        __syncthreads();
        const uint32_t wx = tid & (nwin_pad - 1u);
        const uint32_t px0 = wx * 16u;
        if (px0 < sw) {
            const uint32_t nvalid = min(16u, sw - px0);
            const int j0 = (int)px0;
            for (uint32_t yl = tid >> nwin_log; yl < band_rows; yl += (256u >> nwin_log)) {
                int w[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) w[p] = 0;
                const uint32_t *src = s_row + yl * CAP;
                if (src[0] == 0xffffffffu) {
                    const float cyr = row_cy(yl);
                    for (uint32_t p = 0; p < nvalid; ++p)
                        w[p] = brute_winding(grec, rec_cnt, col_cx((uint32_t)j0 + p), cyr);
                } else {
                    int prev = 0;
                    for (int i = 0; i < CAP; ++i) {
                        const uint32_t en = src[i];
                        const int bq = (int)(en >> 16);
                        const int v = (int)(int16_t)(en & 0xffffu);
#pragma unroll
                        for (int p = 0; p < 16; ++p) {
                            const int j = j0 + p;
                            if (j >= prev && j < bq) w[p] = v;
                        }
                        prev = bq;
                        if (bq == 0xffff) break;
                    }
                }
                const size_t eidx = (out_row0 + yl) * A.out_stride + out_col0 + px0;
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
    }
    STAMP(6);                                   // phase 2: windows + stores
  }   // band loop
}

#ifdef FR_STAMPS
extern "C" int fr_debug_read_stamps(unsigned long long *out8, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
    if (e == hipSuccess && reset) {
        unsigned long long z[8] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z);
    }
    return e == hipSuccess ? 0 : -2;
}
#endif

// LDS plan: padded cx table | region R = max(staged records, window masks / breakpoint
// rows) | s_fill[256] | s_nact[2] | s_cyr[2]
void render_lds_plan(uint32_t strip_w, int n, int mode, uint32_t cap, uint32_t *nwin_log,
                     uint32_t *region, uint32_t *tail, size_t *total)
{
    uint32_t nwin = (strip_w + 15u) / 16u, lg = 0;
    while ((1u << lg) < nwin) ++lg;
    const size_t cx = (((size_t)strip_w * n + 2) * 4 + 15) & ~(size_t)15;
    const size_t a = 256 * sizeof(Rec);
    const size_t b = mode == MODE_COVERAGE_U8 ? (size_t)256 * (8u << lg) : (size_t)256 * cap * 4;
    const size_t t = (cx + (a > b ? a : b) + 15) & ~(size_t)15;
    *nwin_log = lg; *region = (uint32_t)cx; *tail = (uint32_t)t; *total = t + 256 * 4 + 32;
}

template <int MODE, int N, int CAP>
static hipError_t launch_one(RenderArgs a, dim3 grid, hipStream_t stream)
{
    size_t lds;
    render_lds_plan(a.strip_w, N, MODE, CAP, &a.nwin_log, &a.lds_region, &a.lds_tail, &lds);
    auto kern = render_kernel<MODE, N, CAP>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
    return hipGetLastError();
}

template <int MODE, int N>
static hipError_t launch_cap(const RenderArgs &a, dim3 grid, hipStream_t stream)
{
    if (a.kmax <= 8) return launch_one<MODE, N, 8>(a, grid, stream);
    if (a.kmax <= 16) return launch_one<MODE, N, 16>(a, grid, stream);
    return launch_one<MODE, N, 32>(a, grid, stream);
}

hipError_t launch_render(const RenderArgs &a, int mode, int n, hipStream_t stream)
{
    const dim3 grid((uint32_t)((size_t)a.n_jobs * a.band_groups * a.strips));
    if (a.strip_w == 0 || a.strip_w > 256u || (a.strip_w & 15u)) return hipErrorInvalidValue;
    if (mode == MODE_COVERAGE_U8) {
        if (n == 1) return launch_cap<MODE_COVERAGE_U8, 1>(a, grid, stream);
        if (n == 2) return launch_cap<MODE_COVERAGE_U8, 2>(a, grid, stream);
        if (n == 4) return launch_cap<MODE_COVERAGE_U8, 4>(a, grid, stream);
        return hipErrorInvalidValue;
    }
    if (n != 1) return hipErrorInvalidValue;
    if (mode == MODE_WINDING_I16) return launch_cap<MODE_WINDING_I16, 1>(a, grid, stream);
    if (mode == MODE_GRAY_DEBUG) return launch_cap<MODE_GRAY_DEBUG, 1>(a, grid, stream);
    if (mode == MODE_MASK_NONZERO) return launch_cap<MODE_MASK_NONZERO, 1>(a, grid, stream);
    return hipErrorInvalidValue;
}

}  // namespace fr
