// fr_render.hip — the hot path: per-pixel winding / analytic coverage of quadratic
// Bezier contours on gfx950 (wave64, LDS-staged, no MFMA: not a dense contraction).
//
// Replaces the O(W*H*S) loop of renderGlyph + glyphWindingAt
// (/root/reference/src/tools/render_glyph.zig:24-31, :35-73) with a formulation
// that yields the SAME integers:
//
//   winding(cx, cy) = sum over accepted roots k of  sign_k * [ not (xx_k < cx) ]   (:54,:66)
//
// where (xx_k, sign_k) depend only on the segment and the ROW (cy).  Per workgroup = (cell, group of
// wave bands, <= 256-px column strip), 4 waves; a wave band = 64 sample rows (64/N pixel rows):
//   set-up   every thread builds one candidate root record of the glyph straight into LDS, with the
//            EXACT range [ra, re) of this cell's sample rows on which the reference accepts it
//            (fr_records.hpp: the acceptance test is monotone in cy); the padded table of the exact
//            sample abscissae cx(j); ONE workgroup barrier.  Then the waves never meet again.
//   layout   per band, each lane clips the ranges of its 4 records to the band; one DPP prefix sum
//            places every record's run of (record, row) pairs in one sequence (marker at the run's
//            first slot) — no loop over records.
//   evaluate all 64 lanes take consecutive pairs (a DPP max-scan over the markers names the record):
//            t, xx, sign in the reference's own f32 operation order and its three acceptance tests,
//            J = #{sample columns j : cx(j) <= xx} against the table, append (J, step) to the row's
//            list in LDS.
//   sort     one lane per sample row: list -> registers (two 16-bit slots each), packed sorting network.
//   toggles  right to left with the running winding: a slot that changes zero <-> non-zero XORs a prefix
//            mask into the 64-bit LDS word of the 16-pixel window holding it (ds_xor_b64) and flips a
//            per-row "windows to my left are filled" parity word — O(crossings) per row.
//   windows  one lane per 16-pixel window: N mask words -> SWAR popcount per pixel -> 16 output bytes,
//            one 16-B store per lane (256 B per row run).
// Per-pixel work is O(crossings of its row), not O(segments).  A row with more than CAP crossings takes
// the direct sum over records (same integers, slower), inside the kernel.  DESIGN.md §3-§4 has the
// arguments and the measurements.
#include "fr_records.hpp"
#include <cstdio>

namespace fr {

enum { MODE_WINDING_I16 = 0, MODE_GRAY_DEBUG = 1, MODE_MASK_NONZERO = 2, MODE_COVERAGE_U8 = 3 };

__device__ __forceinline__ uint32_t gray_debug(int w)
{
    int v = w * 20 + 100;                       // render_glyph.zig:28
    return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__device__ __forceinline__ float bcast(float v, uint32_t k)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)k));
}

// wave64 inclusive scans on DPP (row_shr within the 16-lane rows, then row_bcast:15 / :31 carry
// the row totals across rows): 6 VALU operations, no LDS.  `old` = 0 is the identity of both.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t x)
{
    x += dpp0<0x111, 0xf>(x);                   // row_shr:1
    x += dpp0<0x112, 0xf>(x);                   // row_shr:2
    x += dpp0<0x114, 0xf>(x);                   // row_shr:4
    x += dpp0<0x118, 0xf>(x);                   // row_shr:8
    x += dpp0<0x142, 0xa>(x);                   // row_bcast:15 -> rows 1, 3
    x += dpp0<0x143, 0xc>(x);                   // row_bcast:31 -> rows 2, 3
    return x;
}
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t x)
{
    x = max(x, dpp0<0x111, 0xf>(x));
    x = max(x, dpp0<0x112, 0xf>(x));
    x = max(x, dpp0<0x114, 0xf>(x));
    x = max(x, dpp0<0x118, 0xf>(x));
    x = max(x, dpp0<0x142, 0xa>(x));
    x = max(x, dpp0<0x143, 0xc>(x));
    return x;
}

#ifndef FR_BAND_PARTS
#define FR_BAND_PARTS 1
#endif
// One crossing = 16 bits: (J << 2) | code, J <= 1024 sample columns of a strip, winding step = code - 1
// (code 2: +1, code 0: -1).  An unused slot is 0xfffd: it sorts last and its step is 0, so the
// suffix sums and the toggle test need no "is this slot used" case.
constexpr uint32_t EMPTY = 0xfffdu;
constexpr uint32_t PCAP = 1024u;               // (record,row) pairs buffered per wave before a dense evaluation round
constexpr uint32_t LSTRIDE = 40u;              // u16 slots per row list: 32 used + pad; an 80-byte
                                               // row stride makes one-row-per-lane b128 reads conflict-free

// Sorting 2H crossings that sit PACKED two per register (d[j] = slot 2j | slot 2j+1 << 16), ascending:
//   1. Batcher's odd-even merge network over the H registers with v_pk_min_u16 / v_pk_max_u16 — the low
//      halves and the high halves are sorted as two independent sequences by the same instructions;
//   2. one "flip" step merges them (low[j] against high[H-1-j]; a half swap, a packed min/max and two
//      byte permutes per register pair): afterwards every low half <= every high half and both are bitonic;
//   3. log2(H) half-cleaner stages, again packed.
// Result: low halves = s[0..H), high halves = s[H..2H).  About half the instructions of the unpacked
// network, no unpacking, half the registers.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pce(uint32_t &a, uint32_t &b)
{
    const u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    a = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(x, y));
    b = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(x, y));
}
template <int H>
__device__ __forceinline__ void packed_sort(uint32_t (&d)[16])
{
#pragma unroll
    for (int p = 1; p < H; p *= 2)
#pragma unroll
        for (int k = p; k >= 1; k /= 2)
#pragma unroll
            for (int j = k % p; j + k < H; j += 2 * k)
#pragma unroll
                for (int i = 0; i < k; ++i)
                    if (i + j + k < H && (i + j) / (2 * p) == (i + j + k) / (2 * p)) pce(d[i + j], d[i + j + k]);
#pragma unroll
    for (int j = 0; j < H / 2; ++j) {
        const uint32_t x = d[j], y = d[H - 1 - j];
        const uint32_t ys = __builtin_amdgcn_alignbit(y, y, 16);                    // halves swapped
        const u16x2 xv = __builtin_bit_cast(u16x2, x), yv = __builtin_bit_cast(u16x2, ys);
        const uint32_t mn = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(xv, yv));
        const uint32_t mx = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(xv, yv));
        d[j] = __builtin_amdgcn_perm(mx, mn, 0x05040100u);                          // min of pair j | max of pair j
        d[H - 1 - j] = __builtin_amdgcn_perm(mx, mn, 0x07060302u);                  // the same of pair H-1-j
    }
#pragma unroll
    for (int k = H / 2; k >= 1; k /= 2)
#pragma unroll
        for (int j = 0; j < H; ++j)
            if (!(j & k)) pce(d[j], d[j + k]);
}
// records staged in LDS per pass
#ifndef FR_RCHUNK
#define FR_RCHUNK 256
#endif
constexpr uint32_t RCHUNK = FR_RCHUNK;
// waves per workgroup: they share one cell's records and cx table and take its wave bands round-robin
#ifndef FR_WG_WAVES
#define FR_WG_WAVES 4
#endif
constexpr uint32_t NW = FR_WG_WAVES;
constexpr uint32_t TAIL_BYTES = (64u / FR_BAND_PARTS) * 4u;   // per-wave LDS tail: fill parity word of every sample row
// Diagnostic build only (make STAMPS=1 -> libfr_raster_stamps.so): per-phase shader-clock
// sums of wave 0 of every workgroup, added to a buffer nothing else reads.  The shipped
// library is built without FR_STAMPS and executes no stamp.
#ifdef FR_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP(i)                                                                          \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        unsigned long long t_;                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                \
        acc_[i] += t_ - t_prev_;                                                          \
        t_prev_ = t_;                                                                     \
    } while (0)
#define STAMP_INIT()                                                                      \
    unsigned long long t_prev_, acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cacc_[4] = {0, 0, 0, 0}; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev_)::"memory")
// one atomic per phase per workgroup (wave 0), at the very end: the stamps themselves stay cheap
#define STAMP_FLUSH()                                                                     \
    do {                                                                                  \
        if (tid == 0)                                                                     \
            for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_stamps[i_], acc_[i_]);            \
        if (lane == 0)                                                                    \
            for (int i_ = 0; i_ < 4; ++i_) atomicAdd(&g_stamps[8 + i_], cacc_[i_]);       \
    } while (0)
#define COUNT(i, n) do { cacc_[(i) - 8] += (unsigned long long)(n); } while (0)
#else
#define STAMP(i) do {} while (0)
#define STAMP_INIT() do {} while (0)
#define STAMP_FLUSH() do {} while (0)
#define COUNT(i, n) do {} while (0)
#endif

// LDS line of sample row r (0..31 of a half band) in a wave's window-mask array.  Swapping
// line parity with bit 2 puts rows r and r+4 (the same sub-row of two adjacent pixel rows,
// read together by one ds_read_b64 in phase 2) into different 128-B halves of the bank space.
#ifdef FR_NO_SWIZZLE
__device__ __forceinline__ uint32_t mask_line(uint32_t r) { return r; }
#else
__device__ __forceinline__ uint32_t mask_line(uint32_t r) { return r ^ ((r >> 2) & 1u); }
#endif
// 64-bit words of padding per mask line: with 0 every line starts in the same LDS bank and the toggles of a
// vertical edge (64 rows, same window) all hit one bank pair; 1 staggers consecutive lines by two banks
#ifndef FR_MASK_PAD
#define FR_MASK_PAD 1
#endif

// LDS hand-off inside ONE wave (writer lanes -> reader lanes of the same wave): LDS operations
// of a wave complete in order, so a drained lgkmcnt plus a compiler barrier is enough — no
// s_barrier, the other three waves of the workgroup are never waited for.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// three workgroups fit a CU's LDS: hold the register allocation to three waves per SIMD (<= 168 VGPRs)
#ifndef FR_WAVES_PER_EU
#define FR_WAVES_PER_EU 3
#endif
#define FR_OCC __attribute__((amdgpu_waves_per_eu(FR_WAVES_PER_EU, FR_WAVES_PER_EU)))
// WLOG >= 0: "uniform" plan — every strip of every job is exactly 16 << WLOG pixels wide and every
// job's height is a multiple of the wave band (RenderArgs::uniform, checked by fr_plan_create): strip
// width, window count and rows per band are compile-time constants (loop counts, addresses, no edge
// cases).  WLOG < 0: the general kernel.
template <int MODE, int N, int CAP, int WLOG>
__global__ __launch_bounds__(64 * FR_WG_WAVES) FR_OCC void render_kernel(const RenderArgs A)
{
    constexpr uint32_t WBAND = 64u / N;         // pixel rows per wave band (64 sample rows)
    constexpr uint32_t PARTS = FR_BAND_PARTS;   // 1: whole band at once (8 KB of masks per wave); 2: two half bands (4 KB)
    constexpr uint32_t PROWS_S = 64u / PARTS;   // sample rows per part
    constexpr uint32_t HROWS = PROWS_S / N;     // pixel rows per part
    constexpr int WCOLS = 16 * N;               // sample columns per 16-pixel window (<= 64)
    constexpr int WSHIFT = (N == 4) ? 6 : (N == 2 ? 5 : 4);
    constexpr unsigned long long WALL = (N == 4) ? ~0ull : ((1ull << WCOLS) - 1ull);
    constexpr bool COV = (MODE == MODE_COVERAGE_U8);
    extern __shared__ __align__(16) unsigned char smem[];

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    STAMP_INIT();
    // workgroup -> (job, band group, strip); one workgroup per cell is the common case (no divisions)
    uint32_t bid = blockIdx.x, strip = 0, bgrp = 0;
    if (A.strips != 1u) { strip = bid % A.strips; bid /= A.strips; }
    if (A.band_groups != 1u) { bgrp = bid % A.band_groups; bid /= A.band_groups; }
    const uint32_t jidx = bid;
    const Job job = A.jobs[jidx];
    const uint32_t x0s = strip * A.strip_w;
    const uint32_t band_first = bgrp * A.bands_per_wg;          // in wave bands
    if (band_first * WBAND >= job.h || x0s >= job.w) return;    // workgroup-uniform
    const uint32_t band_end = min(band_first + A.bands_per_wg, (job.h + WBAND - 1u) / WBAND);
    constexpr bool UNI = WLOG >= 0;
    const uint32_t sw = UNI ? (16u << (UNI ? WLOG : 0)) : min(A.strip_w, job.w - x0s);     // strip width, pixels
    const uint32_t ncol = sw * N;                               // sample columns in the strip
    const int phase = A.phase_center;
    const uint32_t nwin_log = UNI ? (uint32_t)(UNI ? WLOG : 0) : A.nwin_log;   // windows per row, padded to 2^k
    const uint32_t nwin_pad = 1u << nwin_log;
    const uint32_t mrow = nwin_pad + FR_MASK_PAD;                // 64-bit words per mask line

    const uint32_t g = job.glyph;
    const uint32_t seg0 = A.job_seg[2u * (size_t)jidx], nseg = A.job_seg[2u * (size_t)jidx + 1u];   // (loaded with the job)
    const Rec *grec = A.recs + 2u * (size_t)seg0;
    // fused: one slot per candidate root (2 per segment); else the compacted count of prepare_kernel
    // per JOB: a glyph of <= 128 segments has its records built here, in LDS; a larger one is staged from
    // the stand-alone records prepare_kernel rebuilt for it just before this launch (workgroup-uniform)
    const bool fused = A.fused && nseg <= 128u;
    const uint32_t rec_cnt = fused ? 2u * nseg : A.glyph_rec_count[g];

    // LDS: padded cx table | staged records (<= 256, read-only while waves walk them) |
    //      per-wave half-band region (window masks or breakpoint rows) | per-wave fill[32]
    // s_cxp[0] = -inf, s_cxp[1 + j] = cx(j), s_cxp[1 + ncol] = +inf
    float *s_cxp = reinterpret_cast<float *>(smem);
    Rec *s_rec = reinterpret_cast<Rec *>(smem + A.lds_region);                          // [RCHUNK]
    unsigned char *wregion = smem + A.lds_region + A.lds_rec_bytes + (size_t)wave * A.lds_wave_bytes;
    unsigned long long *s_mask = reinterpret_cast<unsigned long long *>(wregion);    // [32][nwin_pad]
    uint32_t *s_row = reinterpret_cast<uint32_t *>(wregion);                         // [32][CAP]
    uint32_t *s_fill = reinterpret_cast<uint32_t *>(smem + A.lds_tail + (size_t)wave * TAIL_BYTES);   // [PROWS_S]

    // Every staged record swaps its bracket [lo, hi] (ray heights) for the EXACT half-open range
    // [ra, re) of this cell's sample rows whose cy lies in it — cy(r) is non-increasing in r, so
    // the range is one interval: ra = min{ r : cy(r) <= hi }, re = min{ r : cy(r) < lo }.  A guess
    // from the affine map is settled against the exact cy(r) (the expression of :27).
    const uint32_t Hs = job.h * (uint32_t)N;    // sample rows of the cell
    auto cy_of = [&](uint32_t r) -> float {
        return ((float)(job.max_y - (int32_t)(r / N)) - sub_off((int)(r % N), N, phase)) / job.scale;
    };
    auto stage = [&](Rec r) {
        uint32_t ra = 1u, re = 0u;
        if (r.lo <= r.hi) {
            const float ph = phase ? 0.5f : 0.0f, top = (float)(Hs - 1u);
            const float ga = ((float)job.max_y - r.hi * job.scale) * (float)N - ph;
            const float ge = ((float)job.max_y - r.lo * job.scale) * (float)N - ph;
            ra = (uint32_t)fminf(fmaxf(ga, 0.0f), top);
            re = (uint32_t)fminf(fmaxf(ge, 0.0f), top);
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            while (ra > 0u && cy_of(ra - 1u) <= r.hi) --ra;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            while (ra < Hs && cy_of(ra) > r.hi) ++ra;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            while (re > 0u && cy_of(re - 1u) < r.lo) --re;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            while (re < Hs && cy_of(re) >= r.lo) ++re;
        }
        r.lo = __builtin_bit_cast(float, ra);
        r.hi = __builtin_bit_cast(float, re);
        return r;
    };
    if (fused) {
        // build my glyph's root records straight into LDS (candidate tid = root tid&1 of segment
        // tid>>1; only glyphs of <= 128 segments come here): no separate
        // prepare pass, no record traffic through HBM.  Slots of non-existent / provably empty
        // roots hold an empty interval and never pass a cull.
        const uint32_t s0g = seg0;
        if (tid < rec_cnt) {
            Rec r;
            RowGeom geo;
            geo.max_y = job.max_y; geo.scale = job.scale; geo.rows = Hs; geo.n = N; geo.phase = phase;
            build_record_rows(A.pts + 2u * (size_t)A.seg_p0[s0g + (tid >> 1)], tid & 1u, geo, r);
            s_rec[tid] = r;
        }
    } else {
        // stage the first (usually only) 256-record chunk: one record per lane, issued first so
        // the latency hides under the cx-table divisions
        if (tid < min(rec_cnt, RCHUNK)) s_rec[tid] = stage(grec[tid]);
    }

    // exact sample abscissae of this strip: cx(j) = (f32(min_x + x) + off(i)) / scale   (:26)
    const int32_t min_xs = job.min_x + (int32_t)x0s;
    auto col_cx = [&](uint32_t j) -> float {
        return ((float)(min_xs + (int32_t)(j / N)) + sub_off((int)(j % N), N, phase)) / job.scale;
    };
    for (uint32_t j = tid; j < ncol; j += 64u * NW) s_cxp[1u + j] = col_cx(j);
    if (tid == 2) s_cxp[0] = -__builtin_inff();
    if (tid == 3) s_cxp[1u + ncol] = __builtin_inff();

    // guess for J from the affine map, fixed up against the exact table
    const float jscale = job.scale * (float)N;
    const float joff = (float)min_xs * (float)N + (phase ? 0.5f : 0.0f) - 1.0f;
    const float ncolf = (float)ncol;
    const size_t out_col0 = (size_t)job.out_x + x0s;
    __syncthreads();                            // the ONLY workgroup barrier for glyphs of <= 256 records
    // a glyph whose records fit one LDS chunk (all but pathological ones): every lane keeps the row
    // ranges of records 4*lane .. 4*lane + 3 in registers for all its bands (consecutive, so that the
    // record index grows along the pair sequence: the marker decode is a max-scan)
    const bool one_chunk = rec_cnt <= RCHUNK;   // workgroup-uniform
    uint32_t rra[4], rre[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t k = 4u * lane + (uint32_t)i;
        const bool have = one_chunk && k < rec_cnt;
        const uint32_t kk = have ? k : 0u;
        rra[i] = have ? __builtin_bit_cast(uint32_t, s_rec[kk].lo) : 1u;
        rre[i] = have ? __builtin_bit_cast(uint32_t, s_rec[kk].hi) : 0u;
    }
    STAMP(0);                                   // setup: job, record staging, cx table

  // wave w takes wave bands band_first + w, + 4, ...; every wave runs the same trip count so the
  // (rare) multi-chunk restaging barriers line up
#if defined(FR_ABLATE) && FR_ABLATE == 20
  if (A.n_jobs != 0xffffffffu) { if (rra[0] + rre[1] + rra[2] + rre[3] == 0x12345u) s_fill[0] = 1u; return; }   // timing-only: set-up alone
#endif
  for (uint32_t band0 = band_first; band0 < band_end; band0 += NW) {
    const uint32_t band = band0 + wave;
    const bool band_valid = band < band_end;
    const uint32_t y0 = band * WBAND;
    const uint32_t nrows = band_valid ? (UNI ? 64u : min(WBAND, job.h - y0) * N) : 0u;    // sample rows of my band

    // ray height of sample row `lane` of the band: cy = (f32(max_y - y) - off(jj)) / scale  (:27)
    const bool row_valid = lane < nrows;
    const uint32_t rr0 = row_valid ? lane : 0u;
    const float cy = ((float)(job.max_y - (int32_t)(y0 + rr0 / N)) - sub_off((int)(rr0 % N), N, phase)) / job.scale;

    // my row's crossings are APPENDED to a wave-private LDS list during the walk (one
    // ds_write_b16 each) and sorted once afterwards; the list lives where the window masks
    // will be (they are built after the list has been pulled into registers)
    uint16_t *s_lists = reinterpret_cast<uint16_t *>(wregion);                         // [64][LSTRIDE]
    uint16_t *mylist = s_lists + lane * LSTRIDE;
    {
        uint4 *f = reinterpret_cast<uint4 *>(mylist);
        const uint4 ones = make_uint4(0xfffdfffdu, 0xfffdfffdu, 0xfffdfffdu, 0xfffdfffdu);
#pragma unroll
        for (uint32_t q = 0; q < CAP / 8u; ++q) f[q] = ones;       // all EMPTY
    }
    uint32_t cnt = 0;
    // dense walk: the wave first lays out every (record, row) pair of the band, then ALL 64 lanes
    // evaluate pairs — a lane works for whichever row its pair names and appends the crossing to
    // that row's list through an LDS counter.  No lane idles while a record is live on only a
    // few rows.
    uint16_t *s_pairs = s_lists + 64u * LSTRIDE;                                       // [PCAP]
    float *s_cy = reinterpret_cast<float *>(s_pairs + PCAP);                          // [64]
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_cy + 64);                         // [64]
    int16_t *s_roff = reinterpret_cast<int16_t *>(s_cnt + 64);                         // [RCHUNK]
    s_cy[lane] = cy;
    s_cnt[lane] = 0u;
    uint32_t npairs = 0;                        // wave-uniform
    // s_pairs holds MARKERS: slot `off` of the pair sequence holds k + 1 where record k's run of
    // pairs starts, 0 elsewhere; a max-scan over the slots recovers every pair's record
    auto zero_markers = [&]() {
        uint4 *mz = reinterpret_cast<uint4 *>(s_pairs);
#pragma unroll
        for (uint32_t q = 0; q < PCAP / 512u; ++q) mz[lane + 64u * q] = make_uint4(0, 0, 0, 0);
        wave_lds_sync();                        // (also orders these 16-B stores before the 2-B marker stores)
    };
    zero_markers();
    const uint32_t row_b0 = band * 64u;         // first sample row of my band (global in the cell)
    auto eval_pairs = [&]() {
        wave_lds_sync();
        STAMP(1);                               // pair layout
        COUNT(9, npairs);
        // one pair per lane: evaluate, find its sample column, append to its row's list.  `k1` = the
        // pair's record + 1 (from the marker scan).  Slots past npairs hold no marker (the buffer is
        // zeroed per band): lanes past the end decode the last record and a row that may lie outside
        // the band — they compute like the others (no divergence) and are kept from the table walk and
        // the append by `live`.
        auto eval_one = [&](uint32_t p, uint32_t k1) {
#if defined(FR_ABLATE) && (FR_ABLATE == 10 || FR_ABLATE == 21)
            const bool live = p < npairs && A.n_jobs == 0xffffffffu;    // timing-only: pairs are collected, never evaluated
            if (live) {
#else
            const bool live = p < npairs;
            {
#endif
                const uint32_t kk = k1 - 1u;
                // (< 64 when live; a lane past the end lands on a row index < 128: still inside s_cy | s_cnt)
                const uint32_t row = (uint32_t)((int32_t)p + (int32_t)s_roff[kk]);
                // (one 24-bit multiply-add for the record's address)
                const Rec r = *reinterpret_cast<const Rec *>(reinterpret_cast<const unsigned char *>(s_rec - 1) + __umul24(k1, (uint32_t)sizeof(Rec)));
                const float cyr = s_cy[row];
                // the reference's operation order, one rounding per operation (:51, :58-61, :53/:65, :67)
                const bool lin = (int32_t)r.flags < 0;                         // REC_LINEAR is the sign bit
                const float delta = cyr * r.a + r.c1 - r.c2;
                const float sq = sqrt_rn(delta);                               // == sqrt(delta), fr_device.hpp
                // B - sqrt == B + (-sqrt) bit for bit: the t- root's record carries the sign bit to flip
                const float sqs = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, sq) ^ r.sqsign);
                const float numq = r.b + sqs, numl = cyr - r.b;
                const float num = lin ? numl : numq;
                const float t = div_by_int(num, lin ? r.c1 : r.a, r.rden);     // == num / d, see fr_device.hpp
                // the row range brackets the accepted set; the reference's own tests decide (:52, :59, :64)
                const bool accepted = (lin || !(delta < 0.0f)) && !(t < 0.0f || t >= 1.0f);
                const float xx = (r.ax * t + r.bx) * t + r.p0x;
                const float dy = r.a * t + (-r.b);
                // winding step code (2: +1, 0: -1): :55 for a line (kept in the flags), :68 for a root
                const uint32_t code = lin ? (r.flags & REC_LIN_PLUS) : ((dy > 0.0f) ? 0u : 2u);
                // J = #{ j in [0, ncol) : cx(j) <= xx }  ==  #{ j : not (xx < cx(j)) }   (:54,:66)
                // <=> s_cxp[J] <= xx < s_cxp[J + 1] in the padded table; guess from the affine
                // map, confirm with one paired read, walk only if the guess is off
                // (the guess is not reference arithmetic: one FMA and one median)
                const float gf = __builtin_amdgcn_fmed3f(__builtin_fmaf(xx, jscale, -joff), 0.0f, ncolf);
                int J = (int)gf;
                {
#if defined(FR_ABLATE) && FR_ABLATE == 11
                    const float c0 = xx, c1 = xx + 1.0f;             // timing-only: no table look-up
#else
                    const float c0 = s_cxp[J], c1 = s_cxp[J + 1];    // one ds_read2_b32
#endif
                    const bool good = (c0 <= xx) & (xx < c1);        // '&': both loads issue together
                    if (!good & live) {
                        while (s_cxp[J + 1] <= xx) ++J;          // +inf sentinel stops it at ncol
                        while (s_cxp[J] > xx) --J;               // -inf sentinel stops it at 0
                    }
                }
#if defined(FR_ABLATE) && FR_ABLATE == 12
                if (live & accepted & (J > 0) && A.n_jobs == 0xffffffffu) {  // timing-only: evaluated, never appended
#else
                if (live & accepted & (J > 0)) {
#endif
                    const uint32_t pos = atomicAdd(&s_cnt[row], 1u);
                    // a row's list has room for CAP slots + 8 of padding: slot 32 + is a dump nobody reads
                    uint16_t *rowlist = s_lists + __umul24(row, LSTRIDE);
                    rowlist[min(pos, 32u)] = (uint16_t)(((uint32_t)J << 2) | code);
                }
            }
        };
        // software pipeline: the marker max-scan of the NEXT 64 pairs is issued before the current 64
        // are evaluated — an independent chain whose DPP steps interleave with the evaluation instead
        // of waiting out the DPP read-after-write gap
        uint32_t k_cur = wave_incl_max((uint32_t)s_pairs[lane]);
        uint32_t carry = (uint32_t)__builtin_amdgcn_readlane((int)k_cur, 63);
        for (uint32_t p0 = 0; p0 < npairs; p0 += 64u) {
            const uint32_t pn = min(p0 + 64u + lane, PCAP - 1u);        // (past the buffer only when the loop ends)
            const uint32_t s_next = wave_incl_max((uint32_t)s_pairs[pn]);
            eval_one(p0 + lane, k_cur);
            k_cur = max(s_next, carry);
            carry = (uint32_t)__builtin_amdgcn_readlane((int)k_cur, 63);
        }
        wave_lds_sync();
        npairs = 0;
        STAMP(2);                               // pair evaluation
    };

    // fast layout: all (<= 256) records in ONE prefix sum — a lane's four runs follow one another
    bool laid_out = false;
    if (one_chunk) {
        uint32_t c[4], r0[4], csum = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r0[i] = max(rra[i], row_b0);
            const uint32_t r1 = min(rre[i], row_b0 + nrows);
            c[i] = r1 > r0[i] ? r1 - r0[i] : 0u;
            csum += c[i];
        }
#if defined(FR_ABLATE) && FR_ABLATE == 8
        csum = (A.n_jobs == 0xffffffffu) ? csum : 0u;                  // timing-only: no walk
#endif
        const uint32_t incl = wave_incl_add(csum);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (tot <= PCAP) {                      // (wave-uniform) else: the generic path below, with flushes
            uint32_t off = incl - csum;
            uint32_t ro[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#if defined(FR_ABLATE) && FR_ABLATE == 8
                if (c[i] && A.n_jobs == 0xffffffffu) {
#else
                if (c[i]) {
#endif
                    s_pairs[off] = (uint16_t)(4u * lane + (uint32_t)i + 1u);
                }
                ro[i] = (r0[i] - row_b0 - off) & 0xffffu;           // row offset of the run (unused if the run is empty)
                off += c[i];
            }
            // my four records' offsets sit side by side: one 8-byte store
            *reinterpret_cast<uint2 *>(s_roff + 4u * lane) = make_uint2(ro[0] | (ro[1] << 16), ro[2] | (ro[3] << 16));
            npairs = tot;
            if (npairs) eval_pairs();
            laid_out = true;
        }
    }

    for (uint32_t base = 0; base < rec_cnt && !laid_out; base += RCHUNK) {
        if (rec_cnt > RCHUNK) {                   // multi-chunk glyph: restage (workgroup-uniform path)
            __syncthreads();
            if (tid < RCHUNK && base + tid < rec_cnt) s_rec[tid] = stage(grec[base + tid]);
            __syncthreads();
        }
#if defined(FR_ABLATE) && FR_ABLATE == 8
        const uint32_t nchunk = (A.n_jobs == 0xffffffffu) ? 1u : 0u;   // timing-only: no walk
#else
        const uint32_t nchunk = min(RCHUNK, rec_cnt - base);
#endif
        for (uint32_t cb = 0; cb < nchunk && nrows; cb += 64u) {
            // lane = record: clip its row range to my band, prefix-sum the run lengths
            const uint32_t k = cb + lane;
            const uint32_t mi = min(k, nchunk - 1u);
            const uint32_t ra = __builtin_bit_cast(uint32_t, s_rec[mi].lo), re = __builtin_bit_cast(uint32_t, s_rec[mi].hi);
            const uint32_t r0 = max(ra, row_b0), r1 = min(re, row_b0 + nrows);
            const uint32_t c = (k < nchunk && r1 > r0) ? r1 - r0 : 0u;
            unsigned long long pending = __builtin_amdgcn_ballot_w64(c != 0u);
            while (pending) {
                const uint32_t cc = ((pending >> lane) & 1ull) ? c : 0u;
                const uint32_t incl = wave_incl_add(cc);
                // runs that still fit the pair buffer (a prefix of the pending ones: incl is monotone)
                const bool fit = (cc != 0u) & (npairs + incl <= PCAP);
                const unsigned long long fm = __builtin_amdgcn_ballot_w64(fit);
                if (fit) {
                    const uint32_t off = npairs + incl - cc;
                    s_pairs[off] = (uint16_t)(k + 1u);
                    s_roff[k] = (int16_t)((int32_t)(r0 - row_b0) - (int32_t)off);
                }
                if (fm) npairs += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63 - (int)__builtin_clzll(fm));
                pending &= ~fm;
                if (pending) { eval_pairs(); zero_markers(); }      // buffer full: evaluate, then go on
            }
        }
        if (npairs) { eval_pairs(); if (base + RCHUNK < rec_cnt) zero_markers(); }   // before the staged records are replaced / the band ends
    }
    cnt = s_cnt[lane];
    COUNT(8, 1);                                // wave bands
    COUNT(11, cnt);                             // crossings of lane 0's row (x64 ~ per band)
    STAMP(1);                                   // pair layout (remainder)
    if (!nrows) continue;                       // this wave has no band in this round (no barriers below)
    if (__ballot(cnt != 0u) == 0ull) {
        // ---- no crossing on any of my 64 sample rows: every winding is 0 — store the band's
        // background directly (cell padding, rows above/below the outline); no lists, no masks
        const uint32_t bg = (MODE == MODE_GRAY_DEBUG) ? 0x64646464u : 0u;      // clamp(0*20+100) = 100 (:28)
        const uint32_t prows_b = nrows / N;
        const uint32_t wx = lane & (nwin_pad - 1u);
        const uint32_t px0 = wx * 16u;
#if defined(FR_ABLATE) && (FR_ABLATE == 21 || FR_ABLATE == 8)
        if (px0 < sw && A.n_jobs == 0xffffffffu) {      // timing-only: no background stores
#else
        if (px0 < sw) {
#endif
            const uint32_t nvalid = min(16u, sw - px0);
            for (uint32_t yl = lane >> nwin_log; yl < prows_b; yl += (64u >> nwin_log)) {
                const size_t eidx = ((size_t)job.out_y + y0 + yl) * A.out_stride + out_col0 + px0;
                if (MODE == MODE_WINDING_I16) {
                    int16_t *dst = reinterpret_cast<int16_t *>(A.out) + eidx;
                    if (nvalid == 16u && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
                        reinterpret_cast<uint4 *>(dst)[0] = make_uint4(0, 0, 0, 0);
                        reinterpret_cast<uint4 *>(dst)[1] = make_uint4(0, 0, 0, 0);
                    } else {
                        for (uint32_t p = 0; p < nvalid; ++p) dst[p] = 0;
                    }
                } else {
                    uint8_t *dst = reinterpret_cast<uint8_t *>(A.out) + eidx;
                    if (nvalid == 16u && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
                        *reinterpret_cast<uint4 *>(dst) = make_uint4(bg, bg, bg, bg);
                    } else {
                        for (uint32_t p = 0; p < nvalid; ++p) dst[p] = (uint8_t)bg;
                    }
                }
            }
        }
        wave_lds_sync();                        // my prefilled list vs. the next band's prefill
        continue;
    }
    // ---- pull my list into registers and sort it by J (network size = the wave's fullest row)
    const bool ovf = cnt > (uint32_t)CAP;
    uint32_t d[16];                             // two slots per register
    uint32_t Hcur, maxcnt;                      // registers in use: sorted slot i sits in d[i % Hcur], half i / Hcur
    {
        const uint4 *f = reinterpret_cast<const uint4 *>(mylist);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 v = make_uint4(0xfffdfffdu, 0xfffdfffdu, 0xfffdfffdu, 0xfffdfffdu);
            if (q < CAP / 8 && (q == 0 || __ballot(cnt > (uint32_t)(8 * q)) != 0ull)) v = f[q];
            d[4 * q + 0] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
        }
        // maxcnt: slots in use in the wave's fullest row, rounded up to the toggle loop's step of 4
        if (CAP > 16 && __ballot(cnt > 16u) != 0ull) {
            packed_sort<16>(d); Hcur = 16u;
            maxcnt = __ballot(cnt > 28u) ? 32u : (__ballot(cnt > 24u) ? 28u : (__ballot(cnt > 20u) ? 24u : 20u));
        } else if (CAP > 8 && __ballot(cnt > 8u) != 0ull) {
            packed_sort<8>(d); Hcur = 8u;
            maxcnt = __ballot(cnt > 12u) ? 16u : 12u;
        } else {
            packed_sort<4>(d); Hcur = 4u;
            maxcnt = __ballot(cnt > 4u) ? 8u : 4u;
        }
    }
    COUNT(10, (unsigned long long)(maxcnt <= 8u) | ((unsigned long long)(maxcnt > 8u && maxcnt <= 12u) << 16) |
                  ((unsigned long long)(maxcnt > 12u && maxcnt <= 16u) << 32) | ((unsigned long long)(maxcnt > 16u) << 48));
    wave_lds_sync();                            // the list region becomes the mask region below
    STAMP(3);                                   // list pull + sort
    // rows with more than CAP crossings (combs, pathological outlines) take the direct sum over the
    // glyph's records — same integers, slow, rare: the coverage modes write the row's window masks from it,
    // the winding-value modes its pixels.
    const unsigned long long ovf_rows = __ballot(ovf);
    // windings of sample row `br` of my band at the 16 sample columns 16 lane .. 16 lane + 15, by the direct
    // sum: every record of the glyph (the stand-alone records in HBM: all of them, whatever sits in LDS) is
    // evaluated once — lane = record, 64 at a time — and broadcast with v_readlane to all lanes
    auto row_windings = [&](uint32_t br, int (&w16)[16]) {
        // fused job: the records are in LDS, each with the exact range of sample rows that accept it;
        // otherwise the stand-alone records in HBM (all of them, whatever chunk sits in LDS), with their brackets
        const uint32_t n_all = fused ? rec_cnt : A.glyph_rec_count[g];
        const uint32_t grow = row_b0 + br;          // sample row in the cell
        const uint32_t col0 = 16u * lane;
        const float cy_r = bcast(cy, br);
#pragma unroll
        for (int c = 0; c < 16; ++c) w16[c] = 0;
        for (uint32_t kb = 0; kb < n_all; kb += 64u) {
            const uint32_t k = kb + lane;
            bool ok = false;
            int J = 0, sgn = 0;
            if (k < n_all) {
                Rec rk;
                bool in_range;
                if (fused) {
                    rk = s_rec[k];
                    in_range = grow >= __builtin_bit_cast(uint32_t, rk.lo) && grow < __builtin_bit_cast(uint32_t, rk.hi);
                } else {
                    rk = grec[k];
                    in_range = cy_r >= rk.lo && cy_r <= rk.hi;
                }
                float xx;
                ok = in_range && rec_cross(rk, cy_r, xx, sgn);
                if (ok) {       // J = #{ j : cx(j) <= xx }, as in the evaluation pass
                    J = (int)__builtin_amdgcn_fmed3f(__builtin_fmaf(xx, jscale, -joff), 0.0f, ncolf);
                    while (s_cxp[J + 1] <= xx) ++J;
                    while (s_cxp[J] > xx) --J;
                }
            }
            unsigned long long m = __ballot(ok && J > 0);
            while (m) {
                const int i = (int)__builtin_ctzll(m);
                m &= m - 1ull;
                const uint32_t sJ = (uint32_t)__builtin_amdgcn_readlane(J, i);
                const int ss = __builtin_amdgcn_readlane(sgn, i);
#pragma unroll
                for (int c = 0; c < 16; ++c) w16[c] += (col0 + (uint32_t)c < sJ) ? ss : 0;
            }
        }
    };

    // ---- phases 1b + 2, one half band (32 sample rows) at a time, wave-private LDS
#pragma unroll
    for (uint32_t half = 0; half < PARTS; ++half) {
        if (half * PROWS_S >= nrows) break;                         // wave-uniform
        const bool mine_half = (PARTS == 1u) || ((lane >> 5) == half);
        const uint32_t hrow = lane & (PROWS_S - 1u);                // my row inside the part
        const uint32_t prow0 = y0 + half * HROWS;                   // first pixel row of the half
        const uint32_t prows = UNI ? HROWS : min(HROWS, job.h - prow0);   // pixel rows in the half
        const size_t out_row0 = (size_t)job.out_y + prow0;
        if (COV) {
            // zero the half's masks cooperatively (consecutive 16-B stores: no bank conflicts)
            {
                uint4 *z = reinterpret_cast<uint4 *>(wregion);
                if (nwin_log == 4u) {           // 256-px strip: a fixed number of stores per lane, no loop
                    constexpr uint32_t NZ = PROWS_S * (16u + FR_MASK_PAD) / 2u;    // 16-byte units
#pragma unroll
                    for (uint32_t q = 0; q < (NZ + 63u) / 64u; ++q)
                        if (NZ % 64u == 0u || lane + 64u * q < NZ) z[lane + 64u * q] = make_uint4(0, 0, 0, 0);
                } else {
                    for (uint32_t q = lane; q < (PROWS_S * mrow + 1u) / 2u; q += 64u) z[q] = make_uint4(0, 0, 0, 0);
                }
            }
            wave_lds_sync();
            // ---- phase 1b: my row's toggles -> window masks (LDS, XOR), fill parity
#if defined(FR_ABLATE) && FR_ABLATE == 2
            if (mine_half && A.n_jobs == 0xffffffffu) {     // timing-only: no toggles
#else
            if (mine_half) {
#endif
                unsigned long long *line = s_mask + mask_line(hrow) * mrow;
                uint32_t fill = 0;
                // right to left: `run` = winding right of the slots handled so far; a slot toggles
                // inside/outside iff the winding changes between zero and non-zero across it.
                // Unused slots step by 0; an over-full row (settled by the direct sum below) never reaches 0.
                int run = ovf ? 0x40000000 : 0;
                bool zero = !ovf;
                auto slot = [&](uint32_t ei) {              // ei: (J << 2) | code in the low 16 bits
                    run += (int)(ei & 3u) - 1;
                    const bool z = run == 0;
                    if (z != zero) {
                        const uint32_t tm1 = ((ei >> 2) & 0x3fffu) - 1u;        // toggle column - 1, 0 .. ncol-1
                        const uint32_t wv = tm1 >> WSHIFT;
                        // columns [0, t) of the window: (tm1 mod WCOLS) + 1 low bits
                        atomicXor(line + wv, ~0ull >> (63u - (tm1 & (uint32_t)(WCOLS - 1))));
                        fill ^= (1u << wv) - 1u;                                // every window to the left flips
                    }
                    zero = z;
                };
                // descending: the high halves hold slots Hcur .. 2 Hcur - 1, the low halves 0 .. Hcur - 1.
                // Registers are visited four at a time — one scalar test per group; a slot past the
                // wave's fullest row is unused and steps by 0.
#pragma unroll
                for (int g = CAP / 8 - 1; g >= 0; --g) {
                    if ((uint32_t)(4 * g) >= Hcur || Hcur + (uint32_t)(4 * g) >= maxcnt) continue;      // wave-uniform
#pragma unroll
                    for (int j = 4 * g + 3; j >= 4 * g; --j) slot(d[j] >> 16);
                }
#pragma unroll
                for (int g = CAP / 8 - 1; g >= 0; --g) {
                    if ((uint32_t)(4 * g) >= Hcur || (uint32_t)(4 * g) >= maxcnt) continue;             // wave-uniform
#pragma unroll
                    for (int j = 4 * g + 3; j >= 4 * g; --j) slot(d[j]);
                }
                s_fill[hrow] = fill;
            }
            wave_lds_sync();
            if (ovf_rows) {
                // ---- over-full sample rows: the toggle loop left their mask lines zero.  For each such
                // row, every record of the glyph (the stand-alone records in HBM: all of them, whatever
                // sits in LDS) is evaluated once — lane = record, 64 at a time — and broadcast with
                // v_readlane to all lanes, each of which keeps the winding of 16 sample columns (lane L:
                // columns 16 L ...).  Non-zero windings become the row's mask bits; its fill parity is 0.
                const uint32_t col0 = 16u * lane;
                unsigned long long todo_rows = (PARTS == 1u) ? ovf_rows : (ovf_rows >> (half * PROWS_S)) & (~0ull >> (64u - PROWS_S));
                while (todo_rows) {
                    const uint32_t r = (uint32_t)__builtin_ctzll(todo_rows);        // row inside the part
                    todo_rows &= todo_rows - 1ull;
                    int w16[16];
                    row_windings(half * PROWS_S + r, w16);
                    uint32_t bits = 0;
#pragma unroll
                    for (int c = 0; c < 16; ++c) bits |= (w16[c] != 0 ? 1u : 0u) << c;
                    // N lanes share a window word of 16 N columns: OR their 16-bit parts together
                    const uint32_t q = lane % (uint32_t)N;
                    uint32_t lo = (q < 2u) ? bits << (16u * q) : 0u, hi = (q >= 2u) ? bits << (16u * (q - 2u)) : 0u;
                    if (N >= 2) { lo |= (uint32_t)__shfl_xor((int)lo, 1); hi |= (uint32_t)__shfl_xor((int)hi, 1); }
                    if (N == 4) { lo |= (uint32_t)__shfl_xor((int)lo, 2); hi |= (uint32_t)__shfl_xor((int)hi, 2); }
                    if (q == 0u && col0 < ncol)
                        s_mask[mask_line(r) * mrow + lane / (uint32_t)N] = ((unsigned long long)hi << 32) | lo;
                }
                wave_lds_sync();
            }
            STAMP(4);                           // phase 1b: zero + toggles (+ over-full rows)

            // ---- phase 2: one lane per 16-pixel window
            // window (yl, wx) = pixels [16 wx, 16 wx + 16) of the half's pixel row yl.  Whether 16-byte
            // stores apply is one wave-uniform fact (windows are 16 bytes apart) plus "the window is full".
            uint8_t *const out_half = reinterpret_cast<uint8_t *>(A.out) + out_row0 * A.out_stride + out_col0;
            auto store_window = [&](uint32_t yl, uint32_t wxx, const uint32_t (&pk)[4]) {
                const uint32_t px = wxx * 16u;
                uint8_t *dst = out_half + (size_t)yl * A.out_stride + px;
#if defined(FR_ABLATE) && FR_ABLATE == 13
                if ((pk[0] ^ pk[1] ^ pk[2] ^ pk[3]) == 0x12345678u) {            // timing-only: windows computed, (almost) never stored
                    *reinterpret_cast<uint4 *>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                }
#else
                if (sw - px >= 16u) {
                    // one 16-byte store, aligned or not (global memory takes unaligned vector stores)
                    const uint4 v = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    __builtin_memcpy(dst, &v, 16);
                } else {
                    const uint32_t nvalid = min(16u, sw - px);
                    for (uint32_t p = 0; p < nvalid; ++p) dst[p] = (uint8_t)(pk[p >> 2] >> (8 * (p & 3)));
                }
#endif
            };
            auto popcount_window = [&](uint32_t yl, uint32_t wxx) {
                unsigned long long mask[N];
#pragma unroll
                for (int rr = 0; rr < N; ++rr) {
                    const uint32_t r = yl * N + rr;
                    const unsigned long long mk = s_mask[mask_line(r) * mrow + wxx];
                    // "windows left of a toggle are filled" parity of this window: 0 or all ones
                    const uint32_t fl = (uint32_t)__builtin_amdgcn_sbfe((int)s_fill[r], wxx, 1u);
                    mask[rr] = mk ^ (((unsigned long long)(fl & (uint32_t)(WALL >> 32)) << 32) | (fl & (uint32_t)WALL));
                }
                uint32_t pk[4];
                if (N == 4) {
                    // per 32-bit half (8 pixels x 4 sample columns x 4 rows): a 4 x 4 BYTE transpose (two rounds
                    // of v_perm_b32) turns the four rows' dwords into four dwords that each hold all 16
                    // samples of two pixels — low nibbles one pixel, high nibbles the next; a masked
                    // v_bcnt_u32_b32 counts a pixel; four counts are packed into a dword and mapped at once
#pragma unroll
                    for (int hlf = 0; hlf < 2; ++hlf) {
                        const uint32_t x0 = (uint32_t)(mask[0] >> (32 * hlf)), x1 = (uint32_t)(mask[1] >> (32 * hlf));
                        const uint32_t x2 = (uint32_t)(mask[2] >> (32 * hlf)), x3 = (uint32_t)(mask[3] >> (32 * hlf));
                        const uint32_t a0 = __builtin_amdgcn_perm(x1, x0, 0x05010400u), a1 = __builtin_amdgcn_perm(x1, x0, 0x07030602u);
                        const uint32_t b0 = __builtin_amdgcn_perm(x3, x2, 0x05010400u), b1 = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
                        uint32_t y[4];
                        y[0] = __builtin_amdgcn_perm(b0, a0, 0x05040100u); y[1] = __builtin_amdgcn_perm(b0, a0, 0x07060302u);
                        y[2] = __builtin_amdgcn_perm(b1, a1, 0x05040100u); y[3] = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
                        uint32_t k[8];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            k[2 * i] = (uint32_t)__popc(y[i] & 0x0f0f0f0fu);             // pixel 2i of the half
                            k[2 * i + 1] = (uint32_t)__popc(y[i] & 0xf0f0f0f0u);         // pixel 2i + 1
                        }
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            // four counts -> four bytes: three byte permutes (selector 0x0c = a zero byte)
                            const uint32_t lo2 = __builtin_amdgcn_perm(k[4 * q + 1], k[4 * q], 0x0c0c0400u);
                            const uint32_t hi2 = __builtin_amdgcn_perm(k[4 * q + 3], k[4 * q + 2], 0x0c0c0400u);
                            const uint32_t p4 = __builtin_amdgcn_perm(hi2, lo2, 0x05040100u);
                            // round_half_up(255*k/16) = 16k - (k > 8), per byte
                            pk[2 * hlf + q] = (p4 << 4) - (((p4 + 0x07070707u) >> 4) & 0x01010101u);
                        }
                    }
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
                store_window(yl, wxx, pk);
            };
            const uint32_t wx = lane & (nwin_pad - 1u);
#if defined(FR_ABLATE) && FR_ABLATE == 3
            const bool win_on = A.n_jobs == 0xffffffffu;    // timing-only: no windows, no stores
#else
            const bool win_on = true;
#endif
            if (wx * 16u < sw && win_on) {
                if (nwin_log == 4u && prows == 16u && N == 4) {
                    // the common full case (256-px strip, 16 pixel rows): four windows per lane, no loop
#pragma unroll
                    for (uint32_t it = 0; it < 4u; ++it) popcount_window((lane >> 4) + 4u * it, wx);
                } else {
                    for (uint32_t yl = lane >> nwin_log; yl < prows; yl += (64u >> nwin_log)) popcount_window(yl, wx);
                }
            }
            wave_lds_sync();                    // masks are re-zeroed by the next half / band
            STAMP(5);                           // phase 2: windows + stores
        } else {
            // ---- winding-value modes (N == 1): breakpoints (b_i, winding on [b_{i-1}, b_i)) per row
            if (mine_half) {
                uint32_t *dst = s_row + hrow * CAP;
                int run = 0;
                auto slot = [&](uint32_t ei, uint32_t i) {
                    ei &= 0xffffu;
                    const bool have = ei != EMPTY;
                    run += (int)(ei & 3u) - 1;
                    dst[i] = have ? (((ei >> 2) << 16) | ((uint32_t)run & 0xffffu)) : 0xffff0000u;
                };
                // sorted slot i sits in d[i % Hcur], half i / Hcur; slots past 2 Hcur are unused
                for (uint32_t i = 2u * Hcur; i < (uint32_t)CAP; ++i) dst[i] = 0xffff0000u;
#pragma unroll
                for (int j = CAP / 2 - 1; j >= 0; --j)
                    if ((uint32_t)j < Hcur) slot(d[j] >> 16, Hcur + (uint32_t)j);
#pragma unroll
                for (int j = CAP / 2 - 1; j >= 0; --j)
                    if ((uint32_t)j < Hcur) slot(d[j], (uint32_t)j);
                if (ovf) dst[0] = 0xffffffffu;                      // (b = 0xffff, v = -1) cannot occur otherwise
            }
            wave_lds_sync();
            const uint32_t wx = lane & (nwin_pad - 1u);
            const uint32_t px0 = wx * 16u;
            if (px0 < sw) {
                const uint32_t nvalid = min(16u, sw - px0);
                const int j0 = (int)px0;
                for (uint32_t yl = lane >> nwin_log; yl < prows; yl += (64u >> nwin_log)) {
                    int w[16];
#pragma unroll
                    for (int p = 0; p < 16; ++p) w[p] = 0;
                    const uint32_t *src = s_row + yl * CAP;
                    if (src[0] == 0xffffffffu) continue;       // over-full row: written by the direct pass below
                    {
                        // entries whose breakpoint lies at or left of my first column say nothing about my
                        // 16 pixels: count them (cheap, wave-uniform trip count), then walk only the few
                        // that reach into the window
                        uint32_t i = 0;
                        for (uint32_t q = 0; q < min(maxcnt, (uint32_t)CAP); ++q) i += ((int)(src[q] >> 16) <= j0) ? 1u : 0u;
                        int prev = j0;
                        while (i < (uint32_t)CAP) {
                            const uint32_t en = src[i];
                            const int bq = (int)(en >> 16);
                            const int v = (int)(int16_t)(en & 0xffffu);
#pragma unroll
                            for (int p = 0; p < 16; ++p) {
                                const int j = j0 + p;
                                if (j >= prev && j < bq) w[p] = v;
                            }
                            prev = bq;
                            ++i;
                            if (bq >= j0 + 16) break;           // (the unused slots hold breakpoint 0xffff, value 0)
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
            // over-full rows (skipped above): the direct sum, one pixel per sample column, stored from here
            unsigned long long todo_rows = (PARTS == 1u) ? ovf_rows : (ovf_rows >> (half * PROWS_S)) & (~0ull >> (64u - PROWS_S));
            while (todo_rows) {
                const uint32_t r = (uint32_t)__builtin_ctzll(todo_rows);            // row inside the part (N == 1: a pixel row)
                todo_rows &= todo_rows - 1ull;
                int w16[16];
                row_windings(half * PROWS_S + r, w16);
                const size_t eidx = (out_row0 + r) * A.out_stride + out_col0 + 16u * lane;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    if (16u * lane + (uint32_t)c >= sw) continue;
                    if (MODE == MODE_WINDING_I16) reinterpret_cast<int16_t *>(A.out)[eidx + c] = (int16_t)w16[c];
                    else reinterpret_cast<uint8_t *>(A.out)[eidx + c] = (uint8_t)((MODE == MODE_GRAY_DEBUG) ? gray_debug(w16[c]) : (w16[c] != 0 ? 255u : 0u));
                }
            }
            wave_lds_sync();
        }
    }   // half band
  }   // band loop
  STAMP_FLUSH();
}

#ifdef FR_STAMPS
extern "C" int fr_debug_read_stamps(unsigned long long *out16, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z);
    }
    return e == hipSuccess ? 0 : -2;
}
#endif

// LDS plan: padded cx table | staged records [RCHUNK] | 4 x per-wave half-band region (window
// masks [32][nwin_pad] u64, or breakpoint rows [32][CAP] u32) | 4 x fill[32]
void render_lds_plan(uint32_t strip_w, int n, int mode, uint32_t cap, uint32_t *nwin_log,
                     uint32_t *region, uint32_t *rec_bytes, uint32_t *wave_bytes, uint32_t *tail, size_t *total)
{
    uint32_t nwin = (strip_w + 15u) / 16u, lg = 0;
    while ((1u << lg) < nwin) ++lg;
    const size_t cx = (((size_t)strip_w * n + 2) * 4 + 15) & ~(size_t)15;
    const size_t prow = 64u / FR_BAND_PARTS;
    size_t wb = mode == MODE_COVERAGE_U8 ? ((prow * ((1u << lg) + FR_MASK_PAD) * 8u + 15u) & ~(size_t)15) : prow * cap * 4;
    const size_t walk = 64u * LSTRIDE * 2u + PCAP * 2u + 64u * 8u + RCHUNK * 2u;   // lists, pairs / markers, cy, counters, run offsets
    if (wb < walk) wb = walk;                                  // the walk's buffers live here first
    const size_t rb = (size_t)RCHUNK * sizeof(Rec);
    const size_t t = cx + rb + NW * wb;
    *rec_bytes = (uint32_t)rb;
    *nwin_log = lg; *region = (uint32_t)cx; *wave_bytes = (uint32_t)wb; *tail = (uint32_t)t;
    *total = t + NW * TAIL_BYTES;
}

template <int MODE, int N, int CAP, int WLOG>
static hipError_t launch_one(RenderArgs a, dim3 grid, hipStream_t stream, char *name, size_t name_cap)
{
    if (name) snprintf(name, name_cap, "fr::render_kernel<%d, %d, %d, %d>", MODE, N, CAP, WLOG);   // as rocprofv3 names the instance
    if (!grid.x) return hipSuccess;                     // (name only)
    size_t lds;
    render_lds_plan(a.strip_w, N, MODE, CAP, &a.nwin_log, &a.lds_region, &a.lds_rec_bytes, &a.lds_wave_bytes, &a.lds_tail, &lds);
    lds += a.lds_pad;
    auto kern = render_kernel<MODE, N, CAP, WLOG>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipSuccess;                                  // (over-full rows are settled inside render_kernel)
}

template <int MODE, int N, int WLOG>
static hipError_t launch_cap(const RenderArgs &a, dim3 grid, hipStream_t stream, char *name, size_t name_cap)
{
    if (a.kmax <= 8) return launch_one<MODE, N, 8, WLOG>(a, grid, stream, name, name_cap);
    if (a.kmax <= 16) return launch_one<MODE, N, 16, WLOG>(a, grid, stream, name, name_cap);
    return launch_one<MODE, N, 32, WLOG>(a, grid, stream, name, name_cap);
}

uint32_t render_wg_waves() { return NW; }

// launch = false: only name the instance (as rocprofv3 prints it) into `name`
hipError_t launch_render(const RenderArgs &a, int mode, int n, hipStream_t stream, bool launch, char *name, size_t name_cap)
{
    const dim3 grid(launch ? (uint32_t)((size_t)a.n_jobs * a.band_groups * a.strips) : 0u);
    if (a.strip_w == 0 || a.strip_w > 256u || (a.strip_w & 15u)) return hipErrorInvalidValue;
    if (mode == MODE_COVERAGE_U8) {
        if (n == 1) return launch_cap<MODE_COVERAGE_U8, 1, -1>(a, grid, stream, name, name_cap);
        if (n == 2) return launch_cap<MODE_COVERAGE_U8, 2, -1>(a, grid, stream, name, name_cap);
        if (n == 4) {
            // uniform plans of 256- / 128-pixel strips (atlas cells) take the specialised instances
            if (a.uniform && a.strip_w == 256u) return launch_cap<MODE_COVERAGE_U8, 4, 4>(a, grid, stream, name, name_cap);
            if (a.uniform && a.strip_w == 128u) return launch_cap<MODE_COVERAGE_U8, 4, 3>(a, grid, stream, name, name_cap);
            return launch_cap<MODE_COVERAGE_U8, 4, -1>(a, grid, stream, name, name_cap);
        }
        return hipErrorInvalidValue;
    }
    if (n != 1) return hipErrorInvalidValue;
    if (mode == MODE_WINDING_I16) return launch_cap<MODE_WINDING_I16, 1, -1>(a, grid, stream, name, name_cap);
    if (mode == MODE_GRAY_DEBUG) return launch_cap<MODE_GRAY_DEBUG, 1, -1>(a, grid, stream, name, name_cap);
    // winding != 0 ? 255 : 0 is exactly the 1-sample coverage (round_half_up(255 k / 1), k in {0, 1})
    if (mode == MODE_MASK_NONZERO) return launch_cap<MODE_COVERAGE_U8, 1, -1>(a, grid, stream, name, name_cap);
    return hipErrorInvalidValue;
}

}  // namespace fr
