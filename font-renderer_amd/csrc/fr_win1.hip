// fr_win1.hip — the reference's own products at one sample per pixel: Image.Winding (int16 winding numbers),
// renderGlyph's gray map clamp(w * 20 + 100) (/root/reference/src/tools/render_glyph.zig:28) and the non-zero mask
// (:29), for cells of any width and height (renderGlyph's own image sizes included, :14-19) of glyphs of <= 768
// segments; the general render_kernel keeps the rest.
//
// Same integers as glyphWindingAt per pixel (:35-73): winding(x) = sum over accepted roots of step [x < J].
// With one sample per pixel there is no inside/outside rule to apply per sample row, so nothing has to be sorted:
// every crossing subtracts its step from ONE byte of a row of winding DIFFERENCES (LDS, ds_add_u32 on the dword
// holding it) straight from the evaluation — w(x) = w(0) - sum over the crossings with J <= x of their step, byte J
// takes - step, and w(0) = the sum of all steps rides in the high half of the row's crossing counter (one more
// ds_add_u32 that the count needs anyway) — and a window lane integrates 16 pixels exactly as cov4_kernel
// integrates coverage (multiply by 0x01010101 per dword, 3-step chain, 4-step DPP scan across the 16 windows of a
// row), in pixel order.  Bytes carry a bias (32 in LDS, 96
// after the scans) and hold |w| <= 31: a row with more than 31 crossings (it alone could leave that range) takes
// the direct path — 16-bit differences, suffix scan, pixels written from there.  No lists, no sort, no toggles:
// per 4 KB of gray output about half the vector instructions of cov4_kernel's 16-sample pixel.
#include "fr_c4.hpp"
#include <cstdio>
#include <type_traits>

namespace fr {

// MODE1_BITS: the non-zero mask as ONE BIT per pixel in job-local bit planes (A.out = the planes' base, A.job_bits[j] =
// the job's first 32-bit word) — the sign the SDF kernel reads (fr_sdf.hip): an eighth of the bytes of the mask, and
// the output itself is then written only once.  Layout: one plane per 256-pixel column of the cell, ceil(w / 256) of
// them one after the other, each h rows of 8 words (bit x % 32 of word (x % 256) / 32 = pixel x) — so the 16 rows of a
// band of a 256-pixel strip are 512 CONTIGUOUS bytes, which leave as 32 lanes x 16 bytes in one store instruction.
// (Measured, WRITE_SIZE per launch for 16.8 MB of bit planes: rows of ceil(w / 32) words written 2 bytes per window
// 76 MB; this layout written 2 bytes per window 49 MB, 16 bytes per every eighth lane 112 MB, whole 512-byte runs 23 MB;
// one plane per 64-pixel column in whole 128-byte lines 76 MB.)
enum { MODE1_WINDING_I16 = 0, MODE1_GRAY_DEBUG = 1, MODE1_MASK = 2, MODE1_BITS = 3 };
enum { W1_ROWS = 16 };

template <int WLOG, int RPL>
struct W1Lds {
    static constexpr uint32_t NCOL = 16u << WLOG;                           // sample columns = pixels of a strip
    static constexpr uint32_t CX = ((NCOL + 2u) * 4u + 15u) & ~15u;
    static constexpr uint32_t RCAP = 64u * RPL;
    static constexpr uint32_t REC = RCAP * (uint32_t)sizeof(Rec40);
    // pairs per round (one row has <= RCAP).  The two-records-per-lane instances take 128 and settle an over-full row in
    // registers (below) instead of in the marker array: that is what lets SIX of their workgroups share a CU at 256-pixel strips
    static constexpr uint32_t PCAP = RCAP > 256u ? RCAP : (RPL == 2 ? 128u : 256u);
    static constexpr bool WD_LDS = RPL != 2;                                // an over-full row's 16-bit differences live in the marker array
    static constexpr uint32_t EROW = NCOL + 16u;
    static constexpr uint32_t E = W1_ROWS * EROW;
    // per wave: E | markers [PCAP] u16 (later: one row of 16-bit differences) | cy [16] | cnt [16] | roff [RCAP] i16
    static constexpr uint32_t OFF_PAIRS = E;
    static constexpr uint32_t OFF_CY = OFF_PAIRS + PCAP * 2u;
    static constexpr uint32_t OFF_CNT = OFF_CY + 64u;
    static constexpr uint32_t OFF_ROFF = OFF_CNT + 64u;
    static constexpr uint32_t WAVE = (OFF_ROFF + RCAP * 2u + 15u) & ~15u;
    static constexpr uint32_t OFF_WAVES = CX + REC;
    static constexpr uint32_t OFF_WCNT = OFF_WAVES + C4_WAVES * WAVE;
    static constexpr uint32_t OFF_CYT = OFF_WCNT + 64u;                     // ray heights of the cell's rows (cells of <= 256 rows)
    static constexpr uint32_t TOTAL = OFF_CYT + 1024u;
};

__device__ __forceinline__ uint32_t w1_gray(int w)
{
    const int v = w * 20 + 100;                     // render_glyph.zig:28
    return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// the first m of a window's 16 pixels (m <= 0: none), ESZ bytes each: a (and b, the second half of 16 int16 values)
template <uint32_t ESZ>
__device__ __forceinline__ void w1_store_clip(unsigned char *dst, uint4 a, uint4 b, int m)
{
    if (m >= 16) {
        if (ESZ == 2u) { __builtin_memcpy(dst, &a, 16); __builtin_memcpy(dst + 16, &b, 16); }
        else c4_store16(dst, a);
        return;
    }
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i >= m) continue;
        if (ESZ == 2u) reinterpret_cast<uint16_t *>(dst)[i] = (uint16_t)(w[i >> 1] >> (16 * (i & 1)));
        else dst[i] = (unsigned char)(w[i >> 2] >> (8 * (i & 3)));
    }
}

// Sign bits of one row of a narrow strip (64 / 128 pixels) or of a row the direct path settled: lane `wx` of the row's
// consecutive lanes holds the 16 bits of its window; four windows are gathered across the lanes and stored as 8 bytes by
// every fourth lane.  Every lane of the wave must call this (the gather reads its neighbours); `row` = the first byte of
// the strip's part of the row in the bit plane, `ok` = the row lies in the cell.
template <int WLOG>
__device__ __forceinline__ void w1_store_bits(unsigned char *row, uint32_t wx, uint32_t bits16, bool ok)
{
    const uint32_t pair = bits16 | ((uint32_t)__shfl_down((int)bits16, 1) << 16);           // windows wx, wx + 1 (even wx)
    const uint32_t d1 = (uint32_t)__shfl_down((int)pair, 2);
    if (ok && (wx & 3u) == 0u) { const uint2 v = make_uint2(pair, d1); __builtin_memcpy(row + 2u * wx, &v, 8); }
}

template <int WLOG, int MODE, int RPL>
__global__ __launch_bounds__(64 * C4_WAVES) __attribute__((amdgpu_waves_per_eu(w1_occ(RPL, WLOG), w1_occ(RPL, WLOG))))
void win1_kernel(const RenderArgs A)
{
    using L = W1Lds<WLOG, RPL>;
    constexpr uint32_t NW = C4_WAVES;
    constexpr uint32_t RCAP = L::RCAP;
    constexpr uint32_t NCOL = L::NCOL;              // pixels of a strip
    constexpr uint32_t NWIN = 1u << WLOG;
    constexpr uint32_t PCAP = L::PCAP;
    constexpr uint32_t ESZ = (MODE == MODE1_WINDING_I16) ? 2u : 1u;
    extern __shared__ __align__(16) unsigned char smem[];

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    // (the wave index as a SCALAR: everything a band derives from it — rows, ray heights' base, the output address — is then
    // scalar arithmetic and the stores take the band's base from SGPRs; `tid >> 6` alone is a vector value to the compiler)
    // (one instance is left exactly as the compiler liked it — six workgroups per CU on 256-pixel strips, the real font's
    // gray maps: every variant of these changes measured 2 - 4 % slower there; see also PLAIN below)
    const uint32_t wave = (WLOG == 4 && RPL == 2) ? (tid >> 6) : (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    uint32_t bid = blockIdx.x, strip = 0, bgrp = 0;
    if (A.strips != 1u) { strip = bid % A.strips; bid /= A.strips; }
    if (A.band_groups != 1u) { bgrp = bid % A.band_groups; bid /= A.band_groups; }
    const uint32_t jidx = bid;
    const Job job = A.jobs[jidx];
    const uint32_t x0s = strip * NCOL;
    const uint32_t band_first = bgrp * A.bands_per_wg;                      // bands of 16 pixel rows
    if (band_first * W1_ROWS >= job.h || x0s >= job.w) return;             // workgroup-uniform
    const uint32_t band_end = min(band_first + A.bands_per_wg, (job.h + W1_ROWS - 1u) / W1_ROWS);
    const uint32_t wlim = min(NCOL, job.w - x0s);                           // pixels of this strip that lie in the cell
    const int phase = A.phase_center;
    const uint32_t seg0 = A.job_seg[2u * (size_t)jidx], nseg = A.job_seg[2u * (size_t)jidx + 1u];

    float *s_cxp = reinterpret_cast<float *>(smem);
    Rec40 *s_rec = reinterpret_cast<Rec40 *>(smem + L::CX);
    unsigned char *wregion = smem + L::OFF_WAVES + (size_t)wave * L::WAVE;
    uint32_t *s_wcnt = reinterpret_cast<uint32_t *>(smem + L::OFF_WCNT);

    const float soff = phase ? 0.5f : 0.0f;
    // cells of up to 256 rows: every row's ray height cy = (f32(max_y - y) - off) / scale (:27) is computed once — the
    // set-up looks at four rows per candidate and every band at 16, each of them a division otherwise
    const float *cyt = nullptr;
    if (job.h <= 256u) {                                                    // (workgroup-uniform)
        float *t = reinterpret_cast<float *>(smem + L::OFF_CYT);
        if (tid < job.h) t[tid] = ((float)(job.max_y - (int32_t)tid) - soff) / job.scale;
        __syncthreads();
        cyt = t;
    }
    const uint32_t rec_cnt_v = c4_setup<NW, RCAP, 1, NCOL>(A, job, seg0, nseg, x0s, phase, s_cxp, s_rec, s_wcnt, reinterpret_cast<uint32_t *>(smem + L::OFF_WAVES), cyt);
    const uint32_t rec_cnt = (WLOG == 4 && RPL == 2) ? rec_cnt_v : (uint32_t)__builtin_amdgcn_readfirstlane((int)rec_cnt_v);   // (workgroup-uniform: a scalar)
    const int32_t min_xs = job.min_x + (int32_t)x0s;
    const float jscale = job.scale;
    const float joff = (float)min_xs + soff - 1.0f;
    const float ncolf = (float)NCOL;
    const bool few = RPL == 2 || rec_cnt <= 128u;   // (RPL == 2: the plan sends only glyphs of <= 128 candidate roots)
    const uint32_t per = few ? 2u : (uint32_t)RPL;
    uint32_t rra[RPL], rre[RPL];
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const uint32_t k = per * lane + (uint32_t)i;
        const bool have = k < rec_cnt && (!few || i < 2);
        const uint32_t f = s_rec[have ? k : 0u].fr;
        rra[i] = have ? (f & 0xfffu) : 1u;
        rre[i] = have ? ((f >> 12) & 0xfffu) : 0u;
    }

    unsigned char *s_E = wregion;
    uint16_t *s_pairs = reinterpret_cast<uint16_t *>(wregion + L::OFF_PAIRS);
    float *s_cy = reinterpret_cast<float *>(wregion + L::OFF_CY);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(wregion + L::OFF_CNT);
    int16_t *s_roff = reinterpret_cast<int16_t *>(wregion + L::OFF_ROFF);
    const uint32_t wx = lane & (NWIN - 1u);

    for (uint32_t band0 = band_first; band0 < band_end; band0 += NW) {
        const uint32_t band = band0 + wave;
        if (band >= band_end) break;               // (no workgroup barrier below: waves are independent)
        const uint32_t y0 = band * W1_ROWS;         // first pixel row = first sample row of my band
        // ray height of row `lane & 15`: cy = (f32(max_y - y) - off) / scale   (:27)
        const float cy = cyt ? cyt[y0 + (lane & 15u)] : ((float)(job.max_y - (int32_t)(y0 + (lane & 15u))) - soff) / job.scale;
        {
            uint4 *z = reinterpret_cast<uint4 *>(s_E);
            const uint4 bias = make_uint4(0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u);
            constexpr uint32_t NZ = L::E / 16u;
#pragma unroll
            for (uint32_t q = 0; q < (NZ + 63u) / 64u; ++q)
                if (NZ % 64u == 0u || lane + 64u * q < NZ) z[lane + 64u * q] = bias;
        }
        if (lane < 16u) { s_cy[lane] = cy; s_cnt[lane] = 0u; }

        // ---- layout + evaluation: rounds of at most PCAP (record, row) pairs over a span of the band's 16 rows
        uint32_t rr0 = 0;
        while (rr0 < W1_ROWS) {
            uint32_t span = W1_ROWS - rr0;
            uint32_t c[RPL], r0[RPL], csum, incl, tot;
            for (;;) {
                const uint32_t lo = y0 + rr0, hi = lo + span;
                csum = 0;
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    r0[i] = max(rra[i], lo);
                    const uint32_t r1 = min(rre[i], hi);
                    c[i] = r1 > r0[i] ? r1 - r0[i] : 0u;
                    csum += c[i];
                }
                incl = c4_wave_incl_add(csum);
                tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (tot <= PCAP || span == 1u) break;
                span >>= 1;
            }
            if (tot) {
#pragma unroll
                for (uint32_t q = 0; q < (PCAP / 8u + 63u) / 64u; ++q)
                    if (PCAP / 8u % 64u == 0u || lane + 64u * q < PCAP / 8u) reinterpret_cast<uint4 *>(s_pairs)[lane + 64u * q] = make_uint4(0, 0, 0, 0);
                c4_wave_lds_sync();
                uint32_t off = incl - csum;
                uint32_t ro[RPL];
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    if (c[i]) s_pairs[off] = (uint16_t)(per * lane + (uint32_t)i + 1u);
                    ro[i] = (r0[i] - y0 - off) & 0xffffu;
                    off += c[i];
                }
                if (few) *reinterpret_cast<uint32_t *>(s_roff + 2u * lane) = ro[0] | (ro[1] << 16);
                else if constexpr (RPL == 4) *reinterpret_cast<uint2 *>(s_roff + 4u * lane) = make_uint2(ro[0] | (ro[1] << 16), ro[2] | (ro[3] << 16));
                else if constexpr (RPL >= 8) {
#pragma unroll
                    for (int q8 = 0; q8 < RPL / 8; ++q8)
                        reinterpret_cast<uint4 *>(s_roff + (uint32_t)RPL * lane)[q8] = make_uint4(ro[8 * q8] | (ro[8 * q8 + 1] << 16), ro[8 * q8 + 2] | (ro[8 * q8 + 3] << 16),
                                                                                                  ro[8 * q8 + 4] | (ro[8 * q8 + 5] << 16), ro[8 * q8 + 6] | (ro[8 * q8 + 7] << 16));
                }
                c4_wave_lds_sync();
                const uint32_t npairs = tot;
                uint32_t k_cur = c4_wave_incl_max((uint32_t)s_pairs[lane]);
                uint32_t carry = (uint32_t)__builtin_amdgcn_readlane((int)k_cur, 63);
                for (uint32_t p0 = 0; p0 < npairs; p0 += 64u) {
                    const uint32_t pn = min(p0 + 64u + lane, PCAP - 1u);
                    const uint32_t s_next = c4_wave_incl_max((uint32_t)s_pairs[pn]);
                    {
                        const uint32_t p = p0 + lane, k1 = k_cur;
                        const bool livep = p < npairs;
                        const uint32_t row = ((uint32_t)((int32_t)p + (int32_t)s_roff[k1 - 1u])) & 15u;
                        const uint32_t raddr = __umul24(k1, (uint32_t)sizeof(Rec40)) + (L::CX - (uint32_t)sizeof(Rec40));
                        const Rec40 r = *reinterpret_cast<const Rec40 *>(smem + raddr);
                        const float cyr = s_cy[row];
                        // the reference's operation order, one rounding per operation (:51, :58-61, :53/:65, :67); the row
                        // range is exactly the set of rows on which the reference accepts this root (fr_records.hpp)
                        const float delta = cyr * r.a + r.c1 - r.c2;
                        const float sq = sqrt_rn(delta);
                        const float numq = r.b + sq * r.sgn, numl = cyr - r.b;
                        const unsigned long long linm = __builtin_amdgcn_sicmp((int32_t)r.fr, 0, 40 /* ICMP_SLT */);
                        const float num = c4_self(linm, numl, numq);
                        const float t = div_by_int(num, r.a, r.rden);
                        const float xx = (r.ax * t + r.bx) * t + r.p0x;
                        const float dy = r.a * t - r.b;
                        const uint32_t cb = (r.fr >> 24) & 3u, zb = (r.fr >> 26) & 3u;
                        const uint32_t code = (dy > 0.0f) ? zb : cb;
                        // J = #{ x in [0, NCOL) : cx(x) <= xx }   (:54, :66)
                        const float gf = __builtin_amdgcn_fmed3f(__builtin_fmaf(xx, jscale, -joff), 0.0f, ncolf);
                        int J = (int)gf;
                        {
                            const float c0 = s_cxp[J], c1 = s_cxp[J + 1];
                            const bool good = (c0 <= xx) & (xx < c1);
                            if (__builtin_expect(!good & livep, 0)) {
                                while (s_cxp[J + 1] <= xx) ++J;
                                while (s_cxp[J] > xx) --J;
                            }
                        }
                        if (livep & (J > 0)) {
                            // w(x) = w(0) - sum of the steps of the crossings with J <= x: byte J of the row takes - step
                            // (J = NCOL lands in the row's pad), the row's counter word the step (high half: w(0)) and the count
                            const uint32_t q = (uint32_t)J;
                            atomicAdd(reinterpret_cast<uint32_t *>(s_E + row * L::EROW + (q & ~3u)), (1u - code) << (8u * (q & 3u)));
                            atomicAdd(&s_cnt[row], 1u + ((code - 1u) << 16));
                        }
                    }
                    k_cur = max(s_next, carry);
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)k_cur, 63);
                }
                c4_wave_lds_sync();
            }
            rr0 += span;
        }
        c4_wave_lds_sync();
        const uint32_t cnt = s_cnt[lane & 15u] & 0xffffu;
        // (bit planes: 32-byte rows, the plane of 256-pixel column x0s / 256 starts h rows after the previous one)
        const size_t row_bytes = (MODE == MODE1_BITS) ? (size_t)32u : (size_t)A.out_stride * ESZ;
        unsigned char *const out_band = (MODE == MODE1_BITS)
            ? reinterpret_cast<unsigned char *>(A.out) + (size_t)A.job_bits[jidx] * 4u + ((size_t)(x0s / 256u) * job.h + y0) * 32u + (x0s % 256u) / 8u
            : reinterpret_cast<unsigned char *>(A.out) + (((size_t)job.out_y + y0) * A.out_stride + job.out_x + x0s) * ESZ;
        // (16 sign bits of window wx go to bytes 2 wx, 2 wx + 1 of the strip's part of the row; a window past the cell's
        // last word still lies inside the row's 32 bytes: written, never read)
        // pixel rows of this band that lie in the cell; `edge`: the band or the strip is cut by the cell's border
        // (wave-uniform — a whole cell never takes the clipped stores)
        const uint32_t hlim = min((uint32_t)W1_ROWS, job.h - y0);
                const bool edge = __builtin_amdgcn_readfirstlane((int)((wlim < NCOL) | (hlim < (uint32_t)W1_ROWS))) != 0;
        if ((__ballot(cnt != 0u) & 0xffffull) == 0ull) {
            // no crossing on any of my 16 rows: every winding is 0
            const uint32_t bg = (MODE == MODE1_GRAY_DEBUG) ? 0x64646464u : 0u;      // clamp(0 * 20 + 100) (:28)
            const uint4 v = make_uint4(bg, bg, bg, bg);
            if (MODE == MODE1_BITS && WLOG == 4) {
                // (a 256-pixel strip: the band's 512 bytes of sign bits in one store instruction, as below)
                if (lane < 32u && (lane >> 1) < hlim) { const uint4 z = make_uint4(0, 0, 0, 0); __builtin_memcpy(out_band + 16u * lane, &z, 16); }
                c4_wave_lds_sync();
                continue;
            }
            for (uint32_t yl = lane >> WLOG; yl < W1_ROWS; yl += (64u >> WLOG)) {
                if (MODE == MODE1_BITS) {
                    w1_store_bits<WLOG>(out_band + (size_t)yl * row_bytes, wx, 0u, yl < hlim);
                    continue;
                }
                unsigned char *dst = out_band + (size_t)yl * row_bytes + 16u * ESZ * wx;
                if (edge) w1_store_clip<ESZ>(dst, v, v, yl < hlim ? (int)wlim - (int)(16u * wx) : 0);
                else if (ESZ == 2u) { __builtin_memcpy(dst, &v, 16); __builtin_memcpy(dst + 16, &v, 16); }
                else c4_store16(dst, v);
            }
            c4_wave_lds_sync();
            continue;
        }
        // rows with more than 31 crossings could leave the byte range: the direct path, pixels stored from here
        const uint32_t ovf_rows = (uint32_t)(__ballot(cnt > 31u) & 0xffffull);
        if (ovf_rows) {
            uint32_t *s_wd = reinterpret_cast<uint32_t *>(s_pairs);          // [NCOL / 2] x two int16 fields, bias 0x4000
            static_assert(!L::WD_LDS || PCAP * 2u >= NCOL * 2u, "no room for the 16-bit winding differences of a row");
            uint32_t todo = ovf_rows;
            while (todo) {
                const uint32_t r = (uint32_t)__builtin_ctz(todo);
                todo &= todo - 1u;
                const float cy_r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cy), (int)r));
                const uint32_t grow = y0 + r;
                // the reference's evaluation of record k at this row -> J (columns left of the crossing) and its step
                auto evaluate = [&](uint32_t k, int &J, uint32_t &step) {
                    const Rec40 rk = s_rec[k];
                    const bool lin = (int32_t)rk.fr < 0;
                    const float delta = cy_r * rk.a + rk.c1 - rk.c2;
                    const float num = lin ? (cy_r - rk.b) : (rk.b + sqrt_rn(delta) * rk.sgn);
                    const float t = div_by_int(num, rk.a, rk.rden);
                    const float xx = (rk.ax * t + rk.bx) * t + rk.p0x;
                    const float dy = rk.a * t - rk.b;
                    step = ((dy > 0.0f) ? ((rk.fr >> 26) & 3u) : ((rk.fr >> 24) & 3u)) - 1u;
                    J = (int)__builtin_amdgcn_fmed3f(__builtin_fmaf(xx, jscale, -joff), 0.0f, ncolf);
                    const float c0 = s_cxp[J], c1 = s_cxp[J + 1];               // one paired read; the guess is nearly always right
                    if (__builtin_expect(!((c0 <= xx) & (xx < c1)), 0)) {
                        while (s_cxp[J + 1] <= xx) ++J;
                        while (s_cxp[J] > xx) --J;
                    }
                };
                int wcol[16];                                               // the winding at my 16 columns
                if constexpr (L::WD_LDS) {
                    for (uint32_t q = lane; q < NCOL * 2u / 16u; q += 64u)
                        reinterpret_cast<uint4 *>(s_wd)[q] = make_uint4(0x40004000u, 0x40004000u, 0x40004000u, 0x40004000u);
                    c4_wave_lds_sync();
#pragma unroll
                    for (int i = 0; i < RPL; ++i) {
                        if (rra[i] <= grow && grow < rre[i]) {
                            int J; uint32_t step;
                            evaluate(per * lane + (uint32_t)i, J, step);
                            if (J > 0) atomicAdd(&s_wd[(uint32_t)(J - 1) >> 1], step << (16u * ((uint32_t)(J - 1) & 1u)));
                        }
                    }
                    c4_wave_lds_sync();
                    int tot = 0;
                    if (16u * lane < NCOL) {
                        const uint4 lo4 = reinterpret_cast<const uint4 *>(s_wd)[2u * lane], hi4 = reinterpret_cast<const uint4 *>(s_wd)[2u * lane + 1u];
                        const uint32_t dws[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
                        for (int cc = 15; cc >= 0; --cc) {
                            tot += (int)((dws[cc >> 1] >> (16 * (cc & 1))) & 0xffffu) - 0x4000;
                            wcol[cc] = tot;
                        }
                    } else {
#pragma unroll
                        for (int cc = 0; cc < 16; ++cc) wcol[cc] = 0;
                    }
                    const uint32_t incl = c4_wave_incl_add((uint32_t)tot);
                    const int right = (int)((uint32_t)__builtin_amdgcn_readlane((int)incl, 63) - incl);
#pragma unroll
                    for (int cc = 0; cc < 16; ++cc) wcol[cc] += right;
                } else {
                    // (two records per lane: glyphs of few crossings per ray — such a row is rare; every crossing is
                    // broadcast to all lanes, w(j) += step [j < J], no LDS)
#pragma unroll
                    for (int cc = 0; cc < 16; ++cc) wcol[cc] = 0;
                    const int col0 = (int)(16u * lane);
#pragma unroll
                    for (int i = 0; i < RPL; ++i) {
                        const bool hit = rra[i] <= grow && grow < rre[i];
                        unsigned long long m = __ballot(hit);
                        if (m == 0ull) continue;
                        int J = 0; uint32_t step = 0u;
                        if (hit) evaluate(per * lane + (uint32_t)i, J, step);
                        while (m) {
                            const int h = (int)__builtin_ctzll(m);
                            m &= m - 1ull;
                            const int n = __builtin_amdgcn_readlane(J, h) - col0;
                            const int sh = __builtin_amdgcn_readlane((int)step, h);
#pragma unroll
                            for (int cc = 0; cc < 16; ++cc) wcol[cc] += (cc < n) ? sh : 0;
                        }
                    }
                }
                if (MODE == MODE1_BITS) {
                    uint32_t bits16 = 0;
#pragma unroll
                    for (int cc = 0; cc < 16; ++cc) bits16 |= (wcol[cc] != 0 ? 1u : 0u) << cc;
                    // (lanes 0 .. NCOL / 16 - 1 are the row's windows in order)
                    w1_store_bits<WLOG>(out_band + (size_t)r * row_bytes, lane, bits16, 16u * lane < NCOL && r < hlim);
                } else if (16u * lane < wlim && r < hlim) {
                    unsigned char *dst = out_band + (size_t)r * row_bytes + 16u * ESZ * lane;
                    const int mlim = (int)wlim - (int)(16u * lane);
#pragma unroll
                    for (int cc = 0; cc < 16; ++cc) {
                        const int w = wcol[cc];
                        if (cc >= mlim) continue;
                        if (MODE == MODE1_WINDING_I16) reinterpret_cast<int16_t *>(dst)[cc] = (int16_t)w;
                        else dst[cc] = (unsigned char)((MODE == MODE1_GRAY_DEBUG) ? w1_gray(w) : (w != 0 ? 255u : 0u));
                    }
                }
                c4_wave_lds_sync();
            }
        }

        // ---- windows: lane = 16 pixels of one pixel row; integrate, map, store
        constexpr uint32_t K1 = 0x01010101u;
        // (one window pass: the hot form stores whole windows; a band or strip cut by the cell's border runs the same
        // arithmetic in a loop of its own — rolled, clipped stores — so that the hot loop stays as small as it was)
        uint32_t band_bits[(W1_ROWS * NWIN) / 64u];                         // (sign-bit mode, 256-pixel strips: my windows' bits, staged below)
        // (the window addresses depend on the lane only, so the compiler computes them once per kernel, keeps them live
        // across the band loop and — in the 80-register instances — spills them; every reload is then a scratch load, and
        // on gfx9 a wave waits for a load with s_waitcnt vmcnt(0), which also waits for every pixel store it has in flight:
        // the stores of a band went out one HBM round trip at a time.  An opaque copy of the lane per band keeps the
        // addresses where they are used: a few integer instructions per window instead.)
        uint32_t lane_w = lane;
#ifndef FR_W1_LAUNDER_ALL
        if constexpr (!(WLOG == 4 && RPL == 2))
#endif
        asm volatile("" : "+v"(lane_w));
        const uint32_t wx = lane_w & (NWIN - 1u);
        // my window of pass 0, in E and in the output (a 32-bit offset from the band's wave-uniform base: 16 rows of < 2^27
        // bytes — fr_plan_render checks the pitch); pass `it` is 64 >> WLOG rows further down: a constant / a scalar away
        const unsigned char *const e_lane = s_E + (lane_w >> WLOG) * L::EROW + 16u * wx;
        const uint32_t out_lane = (lane_w >> WLOG) * (uint32_t)row_bytes + (16u * wx) * ESZ;
        auto window_pass = [&](uint32_t it, auto clipped) {
            constexpr bool edge = decltype(clipped)::value;
            const uint32_t prow = (lane_w >> WLOG) + it * (64u >> WLOG);
            // (the six-workgroup instance of 256-pixel strips keeps the plain addresses: measured 4 % faster that way)
            constexpr bool PLAIN = (WLOG == 4 && RPL == 2);
            const uint4 e = PLAIN ? *reinterpret_cast<const uint4 *>(s_E + prow * L::EROW + 16u * wx)
                                  : *reinterpret_cast<const uint4 *>(e_lane + it * (64u >> WLOG) * L::EROW);
            // bytes 32 + d: inclusive prefix inside each dword gives 32 (i + 1) + sums; then a bias of 64 per byte
            uint32_t x0 = e.x * K1, x1 = e.y * K1, x2 = e.z * K1, x3 = e.w * K1;
            x0 += 0xbfe00020u;                                              // + 32, 0, - 32, - 64 per byte
            x1 = x1 + __builtin_amdgcn_perm(x0, x0, 0x03030303u) - 0x80604020u;
            x2 = x2 + __builtin_amdgcn_perm(x1, x1, 0x03030303u) - 0x80604020u;
            x3 = x3 + __builtin_amdgcn_perm(x2, x2, 0x03030303u) - 0x80604020u;
            const uint32_t T = (x3 >> 24) - 64u;                            // my window's total (signed)
            uint32_t inc = T;
            if (WLOG == 4) {
                inc += c4_dpp0<0x111>(inc);
                inc += c4_dpp0<0x112>(inc);
                inc += c4_dpp0<0x114>(inc);
                inc += c4_dpp0<0x118>(inc);
            } else {
                uint32_t s;
                s = c4_dpp0<0x111>(inc); inc += (wx >= 1u) ? s : 0u;
                s = c4_dpp0<0x112>(inc); inc += (wx >= 2u) ? s : 0u;
                if (WLOG == 3) { s = c4_dpp0<0x114>(inc); inc += (wx >= 4u) ? s : 0u; }
            }
            const int w0 = (int)s_cnt[prow] >> 16;                           // w(0) of my pixel row
            const uint32_t cin = inc - T + 32u + (uint32_t)w0;              // winding entering my window, + 32: in [1, 63]
            const uint32_t cb4 = __builtin_amdgcn_perm(cin, cin, 0x00000000u);
            x0 += cb4; x1 += cb4; x2 += cb4; x3 += cb4;                     // bytes: 96 + w
            const uint32_t p0 = x0, p1 = x1, p2 = x2, p3 = x3;              // (pixel order: byte x of the row is pixel x)
            const bool direct = (ovf_rows >> prow) & 1u;                    // stored by the direct path above
            if (MODE != MODE1_BITS && direct) return;                       // (the sign-bit store gathers across the lanes: all stay)
            unsigned char *dst = PLAIN ? out_band + (size_t)prow * row_bytes + (size_t)(16u * wx) * ESZ
                                       : (out_band + (size_t)(it * (64u >> WLOG)) * row_bytes) + out_lane;
            const int mclip = prow < hlim ? (int)wlim - (int)(16u * wx) : 0;   // (edge only) pixels of my window inside the cell
            if (MODE == MODE1_BITS) {
                // bit 7 of every byte <- (w != 0), then the four bits of a dword gathered by one multiply
                // (2^24 + 2^17 + 2^10 + 2^3 puts bits 0, 8, 16, 24 at 24 .. 27; no two partial products meet)
                auto b4 = [](uint32_t x) -> uint32_t {
                    const uint32_t z = x ^ 0x60606060u;
                    const uint32_t nz = (((z + 0x7f7f7f7fu) | z) & 0x80808080u) >> 7;
                    return (nz * 0x01020408u) >> 24;
                };
                const uint32_t bits16 = (b4(p0) & 15u) | ((b4(p1) & 15u) << 4) | ((b4(p2) & 15u) << 8) | ((b4(p3) & 15u) << 12);
                if (WLOG == 4) band_bits[it] = bits16;                      // (a 256-pixel strip: the whole band leaves at once, below)
                else w1_store_bits<WLOG>(out_band + prow * (uint32_t)row_bytes, wx, bits16, prow < hlim && !direct);
            } else if (MODE == MODE1_MASK) {
                auto m4 = [](uint32_t x) -> uint32_t {
                    const uint32_t z = x ^ 0x60606060u;                     // 0 where w == 0 (bytes < 0x80)
                    const uint32_t nz = ((z + 0x7f7f7f7fu) | z) & 0x80808080u;
                    return nz | (nz - (nz >> 7));                           // 0x80 -> 0xff
                };
                const uint4 v = make_uint4(m4(p0), m4(p1), m4(p2), m4(p3));
                if (edge) w1_store_clip<1u>(dst, v, v, mclip); else c4_store16(dst, v);
            } else if (MODE == MODE1_GRAY_DEBUG) {
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                auto g2 = [](uint32_t h) -> uint32_t {                      // two 16-bit lanes holding 96 + w
                    u16x2 v = __builtin_bit_cast(u16x2, h);
                    v = __builtin_elementwise_sub_sat(v, (u16x2){91, 91});  // max(w, -5) + 5, in [0, 36]
                    v = v * (u16x2){20, 20};                                // max(w * 20 + 100, 0), <= 720: no 16-bit overflow
                    v = __builtin_elementwise_min(v, (u16x2){255, 255});    // (:28)
                    return __builtin_bit_cast(uint32_t, v);
                };
                auto g4 = [&](uint32_t x) -> uint32_t { return g2(x & 0x00ff00ffu) | (g2((x >> 8) & 0x00ff00ffu) << 8); };
                // Nearly every window holds windings in [-5, 2] only (0 and 1, mostly): there the map has nothing to clamp,
                // gray = 20 (w + 5) <= 140 per byte, and one 32-bit multiply does four pixels.  (A byte below 91 wraps or
                // borrows in the subtraction and shows in the high bits tested; the check is exact.)
                const uint32_t y0 = p0 - 0x5b5b5b5bu, y1 = p1 - 0x5b5b5b5bu, y2 = p2 - 0x5b5b5b5bu, y3 = p3 - 0x5b5b5b5bu;
                const bool plain = (((y0 | y1) | (y2 | y3)) & 0xf8f8f8f8u) == 0u;
                uint4 v;
                if (__builtin_expect(__ballot(!plain) == 0ull, 1)) v = make_uint4(y0 * 20u, y1 * 20u, y2 * 20u, y3 * 20u);
                else v = make_uint4(g4(p0), g4(p1), g4(p2), g4(p3));
                if (edge) w1_store_clip<1u>(dst, v, v, mclip); else c4_store16(dst, v);
            } else {
                typedef short i16x2 __attribute__((ext_vector_type(2)));
                auto w2 = [](uint32_t h) -> uint32_t {
                    const i16x2 v = __builtin_bit_cast(i16x2, h) - (i16x2){96, 96};
                    return __builtin_bit_cast(uint32_t, v);
                };
                uint32_t o[8];
                const uint32_t ps[4] = {p0, p1, p2, p3};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t lo = w2(ps[q] & 0x00ff00ffu), hi = w2((ps[q] >> 8) & 0x00ff00ffu);   // pixels 0, 2 | 1, 3
                    o[2 * q] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
                    o[2 * q + 1] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);
                }
                const uint4 va = make_uint4(o[0], o[1], o[2], o[3]), vb = make_uint4(o[4], o[5], o[6], o[7]);
                // (two plain stores: each covers every other 16 bytes of the row, which the streaming hint would send to
                // memory as half-written sectors — measured 2.6 x slower)
                if (edge) w1_store_clip<2u>(dst, va, vb, mclip);
                else { __builtin_memcpy(dst, &va, 16); __builtin_memcpy(dst + 16, &vb, 16); }
            }
        };
        if (__builtin_expect(!edge, 1)) {
#pragma unroll
            for (uint32_t it = 0; it < (W1_ROWS * NWIN) / 64u; ++it) window_pass(it, std::false_type{});
        } else {
#pragma clang loop unroll(disable)
            for (uint32_t it = 0; it < (W1_ROWS * NWIN) / 64u; ++it) window_pass(it, std::true_type{});
        }
        if (MODE == MODE1_BITS && WLOG == 4) {
            // The band's sign bits are 16 rows x 32 bytes = 512 CONTIGUOUS bytes of the strip's bit plane: through LDS (E has
            // been read) they leave as 32 lanes x 16 bytes — four whole 128-byte lines in one store instruction.  (A row the
            // direct path stored keeps its bits: its two lanes skip.)
            c4_wave_lds_sync();
            uint16_t *stage = reinterpret_cast<uint16_t *>(s_E);            // [16 rows][16 windows] u16
#pragma unroll
            for (uint32_t it = 0; it < (W1_ROWS * NWIN) / 64u; ++it) {
                const uint32_t prow = (lane >> WLOG) + it * (64u >> WLOG);
                stage[prow * 16u + wx] = (uint16_t)band_bits[it];
            }
            c4_wave_lds_sync();
            const uint32_t row = lane >> 1;
            if (lane < 32u && row < hlim && !((ovf_rows >> row) & 1u)) {
                const uint4 v = *reinterpret_cast<const uint4 *>(s_E + 16u * lane);
                __builtin_memcpy(out_band + 16u * lane, &v, 16);
            }
        }
        c4_wave_lds_sync();                        // E is re-initialised by the next band
    }
}

uint32_t win1_band_rows() { return W1_ROWS; }

template <int WLOG, int RPL>
static hipError_t win1_launch_mode(const RenderArgs &a, int mode, dim3 grid, hipStream_t stream, char *name, size_t name_cap)
{
    const size_t lds = W1Lds<WLOG, RPL>::TOTAL + a.lds_pad;
    if (name) snprintf(name, name_cap, "fr::win1_kernel<%d, %d, %d>", WLOG, mode, RPL);      // as rocprofv3 names the instance
    if (!grid.x) return hipSuccess;               // (name only)
    auto launch = [&](auto kern) -> hipError_t {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(64 * C4_WAVES), lds, stream, a);
        return hipGetLastError();
    };
    if (mode == MODE1_WINDING_I16) return launch(win1_kernel<WLOG, MODE1_WINDING_I16, RPL>);
    if (mode == MODE1_GRAY_DEBUG) return launch(win1_kernel<WLOG, MODE1_GRAY_DEBUG, RPL>);
    if (mode == MODE1_BITS) return launch(win1_kernel<WLOG, MODE1_BITS, RPL>);
    return launch(win1_kernel<WLOG, MODE1_MASK, RPL>);
}

template <int WLOG>
static hipError_t win1_launch_rpl(const RenderArgs &a, int mode, uint32_t rec_cap, dim3 grid, hipStream_t stream, char *name, size_t name_cap)
{
    if (rec_cap <= 128u) return win1_launch_mode<WLOG, 2>(a, mode, grid, stream, name, name_cap);
    if (rec_cap <= 256u) return win1_launch_mode<WLOG, 4>(a, mode, grid, stream, name, name_cap);
    if (rec_cap > 512u) return win1_launch_mode<WLOG, 16>(a, mode, grid, stream, name, name_cap);
    return win1_launch_mode<WLOG, 8>(a, mode, grid, stream, name, name_cap);
}

// jobs: cells of any size up to 2048 rows (strips of a.strip_w in {64, 128, 256} pixels and bands of 16 rows; the last of
// each may be partial), one sample per pixel, glyphs with <= 384 segments and <= rec_cap possible root records.
// mode: 0 winding_i16, 1 gray_debug, 2 mask, 3 sign bits (one per pixel, job-local bit plane).  launch = false: only name the instance (as rocprofv3 prints it).
hipError_t launch_win1(const RenderArgs &a, int mode, uint32_t rec_cap, hipStream_t stream, bool launch, char *name, size_t name_cap)
{
    const dim3 grid(launch ? (uint32_t)((size_t)a.n_jobs * a.band_groups * a.strips) : 0u);
    if (a.strip_w == 256u) return win1_launch_rpl<4>(a, mode, rec_cap, grid, stream, name, name_cap);
    if (a.strip_w == 128u) return win1_launch_rpl<3>(a, mode, rec_cap, grid, stream, name, name_cap);
    if (a.strip_w == 64u) return win1_launch_rpl<2>(a, mode, rec_cap, grid, stream, name, name_cap);
    return hipErrorInvalidValue;
}

}  // namespace fr
