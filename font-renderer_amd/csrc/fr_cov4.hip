// fr_cov4.hip — the headline instance of the hot path: 16-sample (4 x 4; also 2 x 2) anti-aliased coverage of
// cells of any size, rebuilt around what bounds it on gfx950.  Same integers as render_kernel<COVERAGE_U8, 4> of
// fr_render.hip (and therefore as the reference's glyphWindingAt per sample,
// /root/reference/src/tools/render_glyph.zig:35-73, non-zero fill :29, box filter = the MSAA average
// resolve of VulkanContext.zig:307-313); a plan takes this kernel for every job it fits (any width and height up to
// 2048 sample rows, glyphs of <= 768 segments: fr_api.hip, fast_class) and the general kernel for the rest.
//
// What the round-2 measurements say (tools/ubench/issue_model*.hip, profiles/r02/issue_model*.txt):
// the path is bound by VECTOR-ALU ISSUE TIME — scalar, LDS and branch instructions of one wave hide under
// the vector instructions of the others.  A wave64 vector instruction holds its SIMD for ~4 cycles, except
// a "fast class" (VOP2 add/sub/and/or/xor/lshrrev/mov on VGPRs or literals, v_add_f32 / v_sub_f32 /
// v_mul_f32) that takes ~2 — but only with an EVEN number of waves per SIMD (at 3 waves/SIMD they cost
// 1.8x); v_sqrt_f32 ~7; a VOP2 v_cndmask whose VCC was not written by the instruction right before it
// ~10-19.  So this kernel (1) runs 4-wave workgroups, four per CU = 4 waves per SIMD; (2) turns the
// inside masks + transpose + popcount of the general kernel into signed byte DIFFERENCES of coverage that
// the window lanes integrate with a handful of fast adds (below); (3) keeps selects next to their compares
// or on SGPR masks; (4) drops work whose result is already known (the acceptance tests inside an exact
// row range).
//
// Coverage by integration.  For one sample row the inside set is a union of prefix intervals with signs:
//   inside(j) = sum_k sigma_k [j < J_k],  sigma_k = [w_left != 0] - [w_right != 0]  (non-zero rule)
// over the row's crossings sorted by J (w_right / w_left = winding right / left of crossing k).  The number
// of inside samples of pixel p (4 sample columns) from prefix [0, J) is clamp(J - 4p, 0, 4), i.e.
//   4 sigma + prefix-sum over q <= p of e[q],   e[P] = sigma (f - 4), e[P + 1] = -sigma f,  P = J >> 2, f = J & 3.
// Every toggle therefore adds two small signed numbers to a byte array E[pixel row][pixel] (LDS, ds_add_u32
// on the dword holding the bytes; bytes start at a bias of 16 so the final fields never borrow), the row's
// constant 4 [w(0) != 0] goes to byte 0, and a window lane (16 pixels of one pixel row) integrates: a
// multiply by 0x01010101 per dword, a 3-step chain across its 4 dwords, a 4-step DPP scan across the 16
// windows of the row — then maps 16 counts to bytes at once.  Integer all the way: the result is the same
// count k of inside samples per pixel, u8 = 16 k - [k > 8] = round_half_up(255 k / 16).
#include "fr_c4.hpp"
#include <cstdio>
#include <type_traits>

namespace fr {

typedef unsigned short c4_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void c4_pce(uint32_t &a, uint32_t &b)
{
    const c4_u16x2 x = __builtin_bit_cast(c4_u16x2, a), y = __builtin_bit_cast(c4_u16x2, b);
    a = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(x, y));
    b = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(x, y));
}
// 2H crossings packed two per register, ascending: low halves = s[0..H), high halves = s[H..2H)
// (the packed odd-even merge network of fr_render.hip)
template <int H>
__device__ __forceinline__ void c4_packed_sort(uint32_t (&d)[16])
{
#pragma unroll
    for (int p = 1; p < H; p *= 2)
#pragma unroll
        for (int k = p; k >= 1; k /= 2)
#pragma unroll
            for (int j = k % p; j + k < H; j += 2 * k)
#pragma unroll
                for (int i = 0; i < k; ++i)
                    if (i + j + k < H && (i + j) / (2 * p) == (i + j + k) / (2 * p)) c4_pce(d[i + j], d[i + j + k]);
#pragma unroll
    for (int j = 0; j < H / 2; ++j) {
        const uint32_t x = d[j], y = d[H - 1 - j];
        const uint32_t ys = __builtin_amdgcn_alignbit(y, y, 16);
        const c4_u16x2 xv = __builtin_bit_cast(c4_u16x2, x), yv = __builtin_bit_cast(c4_u16x2, ys);
        const uint32_t mn = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(xv, yv));
        const uint32_t mx = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(xv, yv));
        d[j] = __builtin_amdgcn_perm(mx, mn, 0x05040100u);
        d[H - 1 - j] = __builtin_amdgcn_perm(mx, mn, 0x07060302u);
    }
#pragma unroll
    for (int k = H / 2; k >= 1; k /= 2)
#pragma unroll
        for (int j = 0; j < H; ++j)
            if (!(j & k)) c4_pce(d[j], d[j + k]);
}

#ifdef FR_C4_STATS
__device__ unsigned long long g_c4_stats[16];
#endif
// Timing-only ablation builds (`make ablate4`, tools/c4_ablate.sh: wrong output by construction, never shipped) cut the
// kernel short at the C4_ABL_* points; their bodies live in fr_cov4_ablate.inc, which only those builds include.
#ifdef FR_C4_ABLATE
#include "fr_cov4_ablate.inc"
#else
#define C4_ABL_LAUNCH_ONLY()
#define C4_ABL_JOB_ONLY()
#define C4_ABL_SEGLOAD_ONLY()
#define C4_ABL_SETUP_ONLY()
#define C4_ABL_EVAL_ONLY()
#define C4_ABL_SORT_ONLY()
#define C4_ABL_KEEP(k) true
#define C4_ABL_NODECODE 0
#endif
// LDS plan (bytes): cx table | records | 8 x per-wave region | per-wave counters of the record compaction
template <int WLOG, int RPL, int NS, int CAP>
struct C4Lds {
    static constexpr uint32_t LSTRIDE = c4_lstride(CAP);
    static constexpr uint32_t NCOL = (16u << WLOG) * (uint32_t)NS;          // sample columns of a strip
    static constexpr uint32_t PRB = 64u / (uint32_t)NS;                     // pixel rows of a wave band (64 sample rows)
    static constexpr uint32_t CX = ((NCOL + 2u) * 4u + 15u) & ~15u;         // padded cx table
    static constexpr uint32_t RCAP = 64u * RPL;                             // root records a workgroup keeps
    static constexpr uint32_t REC = RCAP * (uint32_t)sizeof(Rec40);
    static constexpr uint32_t EROW = (16u << WLOG) + 16u;                   // bytes per pixel row of E (one 16-B pad)
    static constexpr uint32_t E = PRB * EROW;
    // walk buffers: lists [64][LSTRIDE] u16 | markers [PCAP] u16 | cy [64] f32 | cnt [64] u32 | roff [256] i16
    static constexpr uint32_t LISTS = 64u * LSTRIDE * 2u;
    static constexpr uint32_t OFF_PAIRS = LISTS;
    static constexpr uint32_t OFF_CY = OFF_PAIRS + C4_PCAP * 2u;
    static constexpr uint32_t OFF_CNT = OFF_CY + 256u;
    static constexpr uint32_t OFF_ROFF = OFF_CNT + 256u;
    static constexpr uint32_t WALK = OFF_ROFF + RCAP * 2u;
    // a row of 16-bit winding differences for an over-full sample row — only where 32 crossings are kept: the instances
    // that keep <= 16 settle such a row (one in 100 000 there) in registers and leave the LDS to two more workgroups per CU
    static constexpr uint32_t WD = (CAP <= 16) ? 0u : NCOL * 2u;
    static constexpr uint32_t WAVE = (WALK > E + WD ? WALK : E + WD);   // (E + an over-full row's 16-bit differences)
    static constexpr uint32_t OFF_WAVES = CX + REC;
    static constexpr uint32_t OFF_WCNT = OFF_WAVES + C4_WAVES * WAVE;
    static constexpr uint32_t TOTAL = OFF_WCNT + 64u;
};

// finished pixels of one 16-pixel window, clipped to the cell: the first m of the 16 bytes (m <= 0: none)
__device__ __forceinline__ void c4_store_clip(unsigned char *dst, uint4 v, int m)
{
    if (m >= 16) { c4_store16(dst, v); return; }
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < m) dst[i] = (unsigned char)(w[i >> 2] >> (8 * (i & 3)));
}

// One workgroup (4 waves) = one cell (or one group of its wave bands, or one strip of it).  The cell may be ragged:
// any width and height (renderGlyph sizes an image to the glyph's own box, render_glyph.zig:14-19) — the last strip
// and the last band are computed whole and their stores are clipped to the cell.
// WLOG: strip width 16 << WLOG pixels (2: 64, 3: 128, 4: 256).  CAP: crossings a sample row keeps (8 / 16 / 32);
// fuller rows take the direct sum over the glyph's records.  RPL: root records per lane, 2, 4 or 8 — a workgroup
// keeps up to 64 RPL records in LDS (128 / 256: four workgroups per CU; 512: three).  NS: samples per pixel axis,
// 4 (16 samples per pixel) or 2 (4): a wave band is 64 sample rows = 64 / NS pixel rows.
template <int WLOG, int CAP, int RPL, int NS>
__global__ __launch_bounds__(64 * C4_WAVES) __attribute__((amdgpu_waves_per_eu(c4_occ(CAP, WLOG, RPL), c4_occ(CAP, WLOG, RPL))))
void cov4_kernel(const RenderArgs A)
{
    using L = C4Lds<WLOG, RPL, NS, CAP>;
    constexpr uint32_t LSTRIDE = L::LSTRIDE;
    static_assert(NS == 4 || NS == 2, "samples per axis");
    constexpr uint32_t RCAP = L::RCAP;
    constexpr uint32_t NW = C4_WAVES;
    constexpr uint32_t SW = 16u << WLOG;            // strip width, pixels
    constexpr uint32_t NCOL = SW * (uint32_t)NS;    // sample columns
    constexpr uint32_t NWIN = 1u << WLOG;           // 16-pixel windows per pixel row
    constexpr uint32_t PRB = L::PRB;                // pixel rows per wave band
    constexpr int LN = (NS == 4) ? 2 : 1;           // log2 NS
    // SPLIT: the instances of many records (glyphs of > 256 segments: wiggly outlines whose over-full rows come dozens to
    // a band) can walk a band's pairs again in two halves of the sample columns — a 64-crossing tier out of two 32-slot
    // passes, no LDS of its own (below) — instead of settling every over-full row by its own direct sum
    constexpr bool SPLIT = RPL >= 8 && CAP == 32;
    extern __shared__ __align__(16) unsigned char smem[];

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    // (the wave index as a SCALAR: everything a band derives from it — rows, ray heights' base, the output address — is then
    // scalar arithmetic and the stores take the band's base from SGPRs; `tid >> 6` alone is a vector value to the compiler)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    uint32_t bid = blockIdx.x, strip = 0, bgrp = 0;
    if (A.strips != 1u) { strip = bid % A.strips; bid /= A.strips; }
    if (A.band_groups != 1u) { bgrp = bid % A.band_groups; bid /= A.band_groups; }
    const uint32_t jidx = bid;
    C4_ABL_LAUNCH_ONLY();
    const Job job = A.jobs[jidx];
    const uint32_t x0s = strip * SW;
    const uint32_t band_first = bgrp * A.bands_per_wg;
    if (band_first * PRB >= job.h || x0s >= job.w) return;                  // workgroup-uniform
    const uint32_t band_end = min(band_first + A.bands_per_wg, (job.h + PRB - 1u) / PRB);
    const uint32_t wlim = min(SW, job.w - x0s);                             // pixels of this strip that lie in the cell
    const int phase = A.phase_center;
    const uint32_t seg0 = A.job_seg[2u * (size_t)jidx], nseg = A.job_seg[2u * (size_t)jidx + 1u];
    C4_ABL_JOB_ONLY();
    C4_ABL_SEGLOAD_ONLY();
    float *s_cxp = reinterpret_cast<float *>(smem);
    Rec40 *s_rec = reinterpret_cast<Rec40 *>(smem + L::CX);
    unsigned char *wregion = smem + L::OFF_WAVES + (size_t)wave * L::WAVE;
    uint32_t *s_wcnt = reinterpret_cast<uint32_t *>(smem + L::OFF_WCNT);

    uint32_t *const s_next_band = s_wcnt + 15;                              // (the set-up's barriers order this store)
    if (tid == 0u) *s_next_band = band_first + NW;
    const uint32_t rec_cnt = c4_setup<NW, RCAP, NS, NCOL>(A, job, seg0, nseg, x0s, phase, s_cxp, s_rec, s_wcnt, reinterpret_cast<uint32_t *>(smem + L::OFF_WAVES));
    const int32_t min_xs = job.min_x + (int32_t)x0s;
    const float jscale = job.scale * (float)NS;
    const float joff = (float)min_xs * (float)NS + (phase ? 0.5f : 0.0f) - 1.0f;
    const float ncolf = (float)NCOL;
    // every lane keeps the row ranges of its records in registers for all its bands: records RPL*lane ...
    // (consecutive, so the record index grows along the pair sequence and the marker decode is a max-scan)
    const bool few = RPL == 2 || rec_cnt <= 128u;   // (RPL == 2: the plan sends only glyphs of <= 128 candidate roots)              // workgroup-uniform: two records per lane are enough
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

    uint16_t *s_lists = reinterpret_cast<uint16_t *>(wregion);
    uint16_t *s_pairs = reinterpret_cast<uint16_t *>(wregion + L::OFF_PAIRS);
    float *s_cy = reinterpret_cast<float *>(wregion + L::OFF_CY);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(wregion + L::OFF_CNT);
    int16_t *s_roff = reinterpret_cast<int16_t *>(wregion + L::OFF_ROFF);
    unsigned char *s_E = wregion;

    C4_ABL_SETUP_ONLY();
    // (no workgroup barrier below: waves are independent.)  Every wave starts on band `wave` of the group and then takes
    // the next band nobody has started yet (one LDS counter): bands differ a lot in cost — the margins above and below the
    // glyph are nearly free — and a workgroup keeps its LDS until its slowest wave is done.  (Measured, same box: C3 - 1.7 %,
    // S = 256 - 6 %, configs[3]'s shard - 1 %; win1_kernel keeps the static round-robin: on its large cells the dealing was
    // worth 0.3 %.)
#if FR_DYN_BANDS
    for (uint32_t band = band_first + wave; band < band_end;
         band = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lane == 0u ? atomicAdd(s_next_band, 1u) : 0u))) {
#else
    for (uint32_t band0 = band_first; band0 < band_end; band0 += NW) {
        const uint32_t band = band0 + wave;
        if (band >= band_end) break;
#endif
        const uint32_t y0 = band * PRB;
        const uint32_t row_b0 = band * 64u;
        // ray height of sample row `lane` of the band: cy = (f32(max_y - y) - off(jj)) / scale  (:27)
        const float cy = ((float)(job.max_y - (int32_t)(y0 + (lane >> LN))) - sub_off((int)(lane & (uint32_t)(NS - 1)), NS, phase)) / job.scale;
        uint16_t *mylist = s_lists + lane * LSTRIDE;
        auto init_lists = [&]() {
            const uint4 ones = make_uint4(0xfffdfffdu, 0xfffdfffdu, 0xfffdfffdu, 0xfffdfffdu);
#pragma unroll
            for (uint32_t q = 0; q < CAP / 8u; ++q) c4_st16<LSTRIDE>(mylist + 8u * q, ones);
            s_cnt[lane] = 0u;
        };
        init_lists();
        s_cy[lane] = cy;

        // ---- layout + evaluation.  The band's (record, row) pairs form ONE sequence, record by record (lane by lane,
        // a lane's records in order); it is walked in chunks of PCAP pairs — only the markers are per chunk, the row
        // offsets and the running record index carry over — 64 pairs per trip, every trip but the last one full.
        uint32_t c[RPL], off0 = 0u, tot;
        {
            uint32_t r0[RPL], csum = 0;
            {
                const uint32_t lo = row_b0, hi = row_b0 + 64u;
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    r0[i] = max(rra[i], lo);
                    const uint32_t r1 = min(rre[i], hi);
                    c[i] = r1 > r0[i] ? r1 - r0[i] : 0u;
                    csum += c[i];
                }
            }
            const uint32_t incl = c4_wave_incl_add(csum);
            tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (tot) {
                off0 = incl - csum;
                {
                    uint32_t off = off0, ro[RPL];
#pragma unroll
                    for (int i = 0; i < RPL; ++i) {
                        ro[i] = (r0[i] - row_b0 - off) & 0xffffu;
                        off += c[i];
                    }
                    // my records' row offsets sit side by side: one store
                    if (few) *reinterpret_cast<uint32_t *>(s_roff + 2u * lane) = ro[0] | (ro[1] << 16);
                    else if constexpr (RPL == 4) *reinterpret_cast<uint2 *>(s_roff + 4u * lane) = make_uint2(ro[0] | (ro[1] << 16), ro[2] | (ro[3] << 16));
                    else if constexpr (RPL >= 8) {
#pragma unroll
                    for (int q8 = 0; q8 < RPL / 8; ++q8)
                        reinterpret_cast<uint4 *>(s_roff + (uint32_t)RPL * lane)[q8] = make_uint4(ro[8 * q8] | (ro[8 * q8 + 1] << 16), ro[8 * q8 + 2] | (ro[8 * q8 + 3] << 16),
                                                                                                  ro[8 * q8 + 4] | (ro[8 * q8 + 5] << 16), ro[8 * q8 + 6] | (ro[8 * q8 + 7] << 16));
                }
                }
            }
        }
        // one walk of the band's pairs; SPLIT instances can walk again keeping only the crossings with J - 1 in
        // [jlo, jlo + jspan) (every instance's first walk keeps J in 1 .. NCOL: all but the ones left of the strip)
        auto eval_pass = [&](uint32_t jlo, uint32_t jspan) {
            (void)jlo; (void)jspan;
            if (tot) {
                const uint32_t npairs = tot;
                uint32_t carry = 0u;               // record index (+ 1) of the last pair walked so far
              for (uint32_t base = 0; base < npairs; base += (uint32_t)C4_PCAP) {
                // markers: slot `off - base` of the chunk holds k + 1 where record k's run starts, 0 elsewhere
                if (C4_PCAP >= 512 || lane < C4_PCAP / 8) reinterpret_cast<uint4 *>(s_pairs)[lane] = make_uint4(0, 0, 0, 0);
                c4_wave_lds_sync();
                {
                    uint32_t off = off0 - base;    // (wraps below the chunk: an unsigned compare takes both ends)
#pragma unroll
                    for (int i = 0; i < RPL; ++i) {
                        if (c[i] && off < (uint32_t)C4_PCAP) s_pairs[off] = (uint16_t)(per * lane + (uint32_t)i + 1u);
                        off += c[i];
                    }
                }
                c4_wave_lds_sync();
                const uint32_t nhere = min(npairs - base, (uint32_t)C4_PCAP);
                // one pair per lane per trip; the marker max-scan of the NEXT 64 pairs is issued before the
                // current 64 are evaluated (an independent chain that fills the evaluation's wait states)
                uint32_t k_cur = max(c4_wave_incl_max((uint32_t)s_pairs[lane]), carry);
                carry = (uint32_t)__builtin_amdgcn_readlane((int)k_cur, 63);
                for (uint32_t p0 = 0; p0 < nhere && C4_ABL_KEEP(3); p0 += 64u) {
                    const uint32_t pn = min(p0 + 64u + lane, (uint32_t)C4_PCAP - 1u);
                    const uint32_t s_next = c4_wave_incl_max((uint32_t)s_pairs[pn]);
                    {
                        const uint32_t p = base + p0 + lane, k1 = k_cur;
                        const bool livep = p < npairs;
                        // (a lane past the end decodes the last record and a row that may lie outside the band:
                        // it computes like the others and is kept from the table walk and the append)
#if C4_ABL_NODECODE
                        const uint32_t row = (p + (k1 & 1u)) & 63u;                                 // timing-only: no dependent decode loads
                        const uint32_t raddr = L::CX + ((lane & 3u) + (A.n_jobs == 0xffffffffu ? k1 : 0u)) * (uint32_t)sizeof(Rec40);
#else
                        const uint32_t row = (uint32_t)((int32_t)p + (int32_t)s_roff[k1 - 1u]);
                        // (one 24-bit multiply-add for the record's LDS address, small offsets for its five 8-byte reads)
                        const uint32_t raddr = __umul24(k1, (uint32_t)sizeof(Rec40)) + (L::CX - (uint32_t)sizeof(Rec40));
#endif
                        const Rec40 r = *reinterpret_cast<const Rec40 *>(smem + raddr);
                        const float cyr = s_cy[row & 63u];
                        // records are ordered quadratic first, linear last, so a trip is nearly always of one kind:
                        // KIND 1 = all quadratic, 2 = all linear (no delta, no square root), 0 = mixed (both + a select)
                        const unsigned long long linm = __builtin_amdgcn_sicmp((int32_t)r.fr, 0, 40 /* ICMP_SLT */);
                        const unsigned long long livem = __ballot(livep);
                        auto body = [&](auto kind) {
                            constexpr int KIND = decltype(kind)::value;
                            // the reference's operation order, one rounding per operation (:51, :58-61, :53/:65, :67);
                            // the row range [ra, re) is exactly the set of rows on which the reference accepts this
                            // root (fr_records.hpp), so its three rejection tests (:52, :59, :64) are not repeated
                            float num;
                            if (KIND == 2) {
                                num = cyr - r.b;
                            } else {
                                const float delta = cyr * r.a + r.c1 - r.c2;
                                const float sq = sqrt_rn(delta);
                                const float numq = r.b + sq * r.sgn;
                                num = (KIND == 1) ? numq : c4_self(linm, cyr - r.b, numq);
                            }
                            const float t = div_by_int(num, r.a, r.rden);
                            const float xx = (r.ax * t + r.bx) * t + r.p0x;
                            uint32_t code;
                            if (KIND == 2) {
                                code = (r.fr >> 24) & 3u;                              // (:55)
                            } else {
                                const float dy = r.a * t - r.b;                        // (:67)
                                if (KIND == 1) code = (dy > 0.0f) ? 0u : 2u;           // (:68)
                                else code = (dy > 0.0f) ? ((r.fr >> 26) & 3u) : ((r.fr >> 24) & 3u);
                            }
                            // J = #{ j in [0, ncol) : cx(j) <= xx }   (:54, :66) — guess, one paired read, rare walk
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
                            bool keep = livep & (J > 0);
                            if constexpr (SPLIT) keep = livep & ((uint32_t)(J - 1) - jlo < jspan);
                            if (keep) {
                                const uint32_t pos = atomicAdd(&s_cnt[row & 63u], 1u);
                                uint16_t *rowlist = s_lists + __umul24(row & 63u, LSTRIDE);
                                rowlist[min(pos, (uint32_t)CAP)] = (uint16_t)(((uint32_t)J << 2) | code);   // (slot CAP: the dump)
                            }
                        };
#if defined(FR_C4_KINDS)
                        // (measured: 87 vector instructions fewer per wave on C3, yet 2 % slower — 58 more branches; off)
                        if ((linm & livem) == 0ull) body(std::integral_constant<int, 1>{});
                        else if ((~linm & livem) == 0ull) body(std::integral_constant<int, 2>{});
                        else body(std::integral_constant<int, 0>{});
#else
                        (void)livem;
                        body(std::integral_constant<int, 0>{});
#endif
                    }
                    k_cur = max(s_next, carry);
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)k_cur, 63);
                }
                c4_wave_lds_sync();
              }
            }
        };
        eval_pass(0u, NCOL);
        C4_ABL_EVAL_ONLY();
        uint32_t cnt = s_cnt[lane];
        uint8_t *const out_band = reinterpret_cast<uint8_t *>(A.out) + ((size_t)job.out_y + y0) * A.out_stride + job.out_x + x0s;
        const uint32_t wx = lane & (NWIN - 1u);
        // pixel rows of this band that lie in the cell; `edge`: the band or the strip is cut by the cell's border
        // (wave-uniform — a whole cell never takes the clipped stores)
        const uint32_t hlim = min(PRB, job.h - y0);
                const bool edge = __builtin_amdgcn_readfirstlane((int)((wlim < SW) | (hlim < PRB))) != 0;
        if (__ballot(cnt != 0u) == 0ull) {
            // no crossing on any of my 64 sample rows: every winding is 0 — store the band's background
            for (uint32_t yl = lane >> WLOG; yl < PRB; yl += (64u >> WLOG)) {
                const uint4 z = make_uint4(0, 0, 0, 0);
                unsigned char *dst = out_band + (size_t)yl * A.out_stride + 16u * wx;
                if (!edge) c4_store16(dst, z);
                else c4_store_clip(dst, z, yl < hlim ? (int)wlim - (int)(16u * wx) : 0);
            }
            c4_wave_lds_sync();
            continue;
        }
        // ---- pull my list into registers and sort it by J (network size = the wave's fullest row)
        auto pull_sort = [&](uint32_t (&dd)[16], uint32_t n, bool blank, uint32_t &H, uint32_t &mx) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint4 v = make_uint4(0xfffdfffdu, 0xfffdfffdu, 0xfffdfffdu, 0xfffdfffdu);
                if (q < CAP / 8 && (q == 0 || __ballot(n > (uint32_t)(8 * q)) != 0ull)) v = c4_ld16<LSTRIDE>(mylist + 8 * q);
                dd[4 * q + 0] = v.x; dd[4 * q + 1] = v.y; dd[4 * q + 2] = v.z; dd[4 * q + 3] = v.w;
            }
            if (__ballot(blank)) {
                // over-full rows are settled by the direct sum below: their lists must add nothing
#pragma unroll
                for (int q = 0; q < 16; ++q) dd[q] = blank ? 0xfffdfffdu : dd[q];
            }
            if (CAP > 16 && __ballot(n > 16u && !blank) != 0ull) {
                c4_packed_sort<16>(dd); H = 16u;
                mx = __ballot(n > 28u) ? 32u : (__ballot(n > 24u) ? 28u : (__ballot(n > 20u) ? 24u : 20u));
            } else if (CAP > 8 && __ballot(n > 8u && !blank) != 0ull) {
                c4_packed_sort<8>(dd); H = 8u;
                mx = __ballot(n > 12u) ? 16u : 12u;
            } else {
                c4_packed_sort<4>(dd); H = 4u;
                mx = __ballot(n > 4u) ? 8u : 4u;
            }
        };
        bool ovf = cnt > (uint32_t)CAP;
        uint32_t d[16];
        uint32_t Hcur, maxcnt;
        // SPLIT: the crossings left of the strip's middle, when the band was walked in two halves
        [[maybe_unused]] uint32_t dl[SPLIT ? 16 : 1];
        [[maybe_unused]] uint32_t Hl = 0u, maxl = 0u;
        [[maybe_unused]] bool split = false;
        if constexpr (SPLIT) {
            // Over-full rows come in bands (a wiggly outline crosses dozens of neighbouring rows 40 - 60 times): with three
            // or more of them the wave walks the band's pairs twice more, once keeping the crossings of the right half of
            // the sample columns and once those of the left half — 32 slots each, the right half pulled into registers
            // before the left half reuses the lists — and the toggle walk below runs through one after the other (every
            // crossing of the right half lies right of every one of the left).  A row that overflows a half still takes
            // the direct sum.  Three walks instead of one, against ~1 100 vector instructions per over-full row.
#ifdef FR_C4_STATS
            const int n0 = __popcll(__ballot(ovf));
            if (lane == 0) {
                if (n0) { atomicAdd(&g_c4_stats[9], 1ull); atomicAdd(&g_c4_stats[10], (unsigned long long)n0); }   // bands with over-full rows / such rows, first walk
                if (n0 >= 3) { atomicAdd(&g_c4_stats[11], 1ull); atomicAdd(&g_c4_stats[12], (unsigned long long)n0); }   // bands walked in halves / their rows
            }
#endif
            if (__popcll(__ballot(ovf)) >= 3) {
                split = true;
                init_lists();
                eval_pass(NCOL / 2u, NCOL / 2u);             // J in (NCOL / 2, NCOL]
                const uint32_t cr = s_cnt[lane];
                pull_sort(d, cr, cr > (uint32_t)CAP, Hcur, maxcnt);
                c4_wave_lds_sync();
                init_lists();
                eval_pass(0u, NCOL / 2u);                    // J in [1, NCOL / 2]
                const uint32_t cl = s_cnt[lane];
                pull_sort(dl, cl, cl > (uint32_t)CAP, Hl, maxl);
                ovf = (cr > (uint32_t)CAP) | (cl > (uint32_t)CAP);
                if (__ballot(ovf)) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) { d[q] = ovf ? 0xfffdfffdu : d[q]; dl[q] = ovf ? 0xfffdfffdu : dl[q]; }
                }
                cnt = cr + cl;
            }
        }
        if (!split) pull_sort(d, cnt, ovf, Hcur, maxcnt);
        const unsigned long long ovf_rows = __ballot(ovf);
        c4_wave_lds_sync();                        // the list region becomes E below
#ifdef FR_C4_STATS
        // diagnostic build only (make variant NAME=c4stats DEFS=-DFR_C4_STATS): per wave band — sort tier, crossings, over-full rows
        if (lane == 0) {
            atomicAdd(&g_c4_stats[0], 1ull);                                                   // wave bands with crossings
            atomicAdd(&g_c4_stats[Hcur == 4u ? 1 : (Hcur == 8u ? 2 : 3)], 1ull);               // 8- / 16- / 32-slot sort
            atomicAdd(&g_c4_stats[4], (unsigned long long)__popcll(ovf_rows));                 // over-full sample rows
        }
        {
            const uint32_t csum = c4_wave_incl_add(cnt);
            if (lane == 63) atomicAdd(&g_c4_stats[5], (unsigned long long)csum);               // crossings kept (J > 0)
            const unsigned long long g8 = __ballot(cnt > 8u), g16 = __ballot(cnt > 16u), g32 = __ballot(cnt > 32u);
            if (lane == 0) {
                atomicAdd(&g_c4_stats[6], (unsigned long long)__popcll(g8));                   // rows with > 8 / > 16 / > 32 crossings
                atomicAdd(&g_c4_stats[7], (unsigned long long)__popcll(g16));
                atomicAdd(&g_c4_stats[8], (unsigned long long)__popcll(g32));
            }
        }
#endif

        C4_ABL_SORT_ONLY();
        // ---- E: every byte starts at the bias 16
        {
            uint4 *z = reinterpret_cast<uint4 *>(s_E);
            const uint4 bias = make_uint4(0x10101010u, 0x10101010u, 0x10101010u, 0x10101010u);
            constexpr uint32_t NZ = L::E / 16u;
#pragma unroll
            for (uint32_t q = 0; q < (NZ + 63u) / 64u; ++q)
                if (NZ % 64u == 0u || lane + 64u * q < NZ) z[lane + 64u * q] = bias;
        }
        c4_wave_lds_sync();
        // ---- toggles: right to left with the running winding; a crossing that changes zero <-> non-zero adds
        // its two differences to my pixel row's bytes
        if (C4_ABL_KEEP(2)) {
            unsigned char *erow = s_E + (lane >> LN) * L::EROW;
            int run = 0;
            bool zero = true;
            // toggles alternate (zero <-> non-zero), so -sigma of the next toggle is a register that flips:
            // A = 255 * (-sigma), B = NS * (-sigma); V = e0 + 256 e1 = -sigma (255 f + NS) = f A + B
            int A255 = -255, B4 = -NS, ns = -1;
            auto slot = [&](uint32_t dw, int o) {       // crossing (J << 2) | code in bits o .. o + 15 of dw
                run += (int)((dw >> o) & 3u) - 1;
                const bool z = run == 0;
                if (z != zero) {
                    const uint32_t nib = (dw >> (o + 2)) & (uint32_t)(4 * NS - 1);   // (P & 3) << LN | f,  P = J / NS, f = J % NS
                    const int V = __mul24((int)(nib & (uint32_t)(NS - 1)), A255) + B4;
                    const uint32_t sh = (dw >> (o + LN - 1)) & 0x18u;       // 8 (P & 3)
                    uint32_t *dwp = reinterpret_cast<uint32_t *>(erow + ((dw >> (o + LN + 2)) & 0xffcu));
                    atomicAdd(dwp, (uint32_t)V << sh);
                    // the second difference of a pixel in byte 3 belongs to the next dword: e1 = -sigma f
                    // there, 0 elsewhere — f if (P & 3) == 3, i.e. nib - 3 NS saturated at 0
                    const uint32_t g3 = __builtin_elementwise_sub_sat(nib, (uint32_t)(3 * NS));
                    atomicAdd(dwp + 1, (uint32_t)__mul24((int)g3, ns));
                    A255 = -A255; B4 = -B4; ns = -ns;
                }
                zero = z;
            };
            auto walk = [&](const uint32_t (&dd)[16], uint32_t H, uint32_t mx) {
#pragma unroll
                for (int gq = CAP / 8 - 1; gq >= 0; --gq) {
                    if ((uint32_t)(4 * gq) >= H || H + (uint32_t)(4 * gq) >= mx) continue;           // wave-uniform
#pragma unroll
                    for (int j = 4 * gq + 3; j >= 4 * gq; --j) slot(dd[j], 16);
                }
#pragma unroll
                for (int gq = CAP / 8 - 1; gq >= 0; --gq) {
                    if ((uint32_t)(4 * gq) >= H || (uint32_t)(4 * gq) >= mx) continue;               // wave-uniform
#pragma unroll
                    for (int j = 4 * gq + 3; j >= 4 * gq; --j) slot(dd[j], 0);
                }
            };
            walk(d, Hcur, maxcnt);
            if constexpr (SPLIT) { if (split) walk(dl, Hl, maxl); }
            // the row's constant: NS [w(0) != 0], into byte 0 of the pixel row
            if (run != 0) atomicAdd(reinterpret_cast<uint32_t *>(erow), (uint32_t)NS);
        }
        if (__builtin_expect(ovf_rows != 0ull, 0)) {
            // ---- over-full sample rows (more than CAP crossings): the direct sum.  Every record whose row range holds
            // the row is evaluated once more; the winding of every sample column follows from the (J, step) pairs, and
            // lane L turns the non-zero counts of its 16 columns' pixels into E's difference form.
            // Instances that keep 32 crossings (WD != 0): the steps go to a row of 16-bit winding DIFFERENCES in LDS
            // (w(j) = sum over i >= j of d[i]: d[J - 1] += step; the 2 KB next to E, free since the lists were pulled), a
            // suffix sum (16 columns per lane + one wave scan) gives the windings.  Instances that keep <= 16 (WD == 0:
            // glyphs of few crossings per ray — such a row is one in 100 000 there) have no LDS for that row — it is what
            // lets six of their workgroups share a CU — and broadcast every pair to all lanes instead (v_readlane).
            uint32_t *s_wd = reinterpret_cast<uint32_t *>(wregion + L::E);      // [NCOL / 2] x two int16 fields, bias 0x4000
            static_assert(L::WD == 0u || L::WAVE >= L::E + NCOL * 2u, "no room for the winding differences of an over-full row");
            // the reference's evaluation of record k at ray height cy_r -> J (sample columns left of the crossing) and its step
            auto evaluate = [&](uint32_t k, float cy_r, int &J, uint32_t &step) {
                const Rec40 rk = s_rec[k];
                const bool lin = (int32_t)rk.fr < 0;
                const float delta = cy_r * rk.a + rk.c1 - rk.c2;
                const float num = lin ? (cy_r - rk.b) : (rk.b + sqrt_rn(delta) * rk.sgn);
                const float t = div_by_int(num, rk.a, rk.rden);
                const float xx = (rk.ax * t + rk.bx) * t + rk.p0x;
                const float dy = rk.a * t - rk.b;
                step = ((dy > 0.0f) ? ((rk.fr >> 26) & 3u) : ((rk.fr >> 24) & 3u)) - 1u;   // +1 or -1 (mod 2^32)
                J = (int)__builtin_amdgcn_fmed3f(__builtin_fmaf(xx, jscale, -joff), 0.0f, ncolf);
                const float c0 = s_cxp[J], c1 = s_cxp[J + 1];               // one paired read; the guess is nearly always right
                if (__builtin_expect(!((c0 <= xx) & (xx < c1)), 0)) {
                    while (s_cxp[J + 1] <= xx) ++J;
                    while (s_cxp[J] > xx) --J;
                }
            };
            unsigned long long todo = ovf_rows;
            while (todo) {
                const uint32_t r = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1ull;
                const float cy_r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cy), (int)r));
                const uint32_t grow = row_b0 + r;
                unsigned long long hm[RPL];
                uint32_t total = 0u;
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    hm[i] = __ballot(rra[i] <= grow && grow < rre[i]);      // the rows that accept this root (exact)
                    total += (uint32_t)__popcll(hm[i]);
                }
                int wcol[16];                                               // the winding at my 16 sample columns
                if constexpr (L::WD != 0u) {
                    for (uint32_t q = lane; q < NCOL * 2u / 16u; q += 64u)
                        reinterpret_cast<uint4 *>(s_wd)[q] = make_uint4(0x40004000u, 0x40004000u, 0x40004000u, 0x40004000u);
                    c4_wave_lds_sync();
                    auto add = [&](uint32_t k) {
                        int J; uint32_t step;
                        evaluate(k, cy_r, J, step);
                        if (J > 0) atomicAdd(&s_wd[(uint32_t)(J - 1) >> 1], step << (16u * ((uint32_t)(J - 1) & 1u)));
                    };
                    // The records that hold the row are a few dozen of up to 64 RPL, scattered over the lanes' RPL slots:
                    // walking the slots would run RPL divergent evaluations at a few per cent of the lanes each.  Instead
                    // every hit is pushed to a dense lane first (ds_permute: a forward permutation, no LDS memory) — slot
                    // i's hits go to the dense positions fill .. fill + c - 1, its other lanes fill the rest of the same
                    // permutation — and the evaluation runs once per 64 hits.
                    if (__builtin_expect(total <= 128u, 1)) {
                        uint32_t fill = 0u;                                 // dense positions in use
                        uint32_t kd0 = 0xffffffffu, kd1 = 0xffffffffu;      // my dense record index: positions 0 .. 63 / 64 .. 127
#pragma unroll
                        for (int i = 0; i < RPL; ++i) {
                            if (hm[i] == 0ull) continue;                    // (wave-uniform)
                            const bool hit = (hm[i] >> lane) & 1ull;
                            const uint32_t c = (uint32_t)__popcll(hm[i]);
                            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm[i], 0u));
                            // hits -> dense positions fill + rank (lane = position mod 64), the other lanes -> the lanes left over
                            const uint32_t dst = hit ? ((fill + below) & 63u) : ((fill + c + (lane - below)) & 63u);
                            const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)(hit ? per * lane + (uint32_t)i : 0xffffffffu));
                            const uint32_t end = fill + c;
                            kd0 = (lane >= fill && lane < end) ? got : kd0;
                            kd1 = (lane + 64u >= fill && lane + 64u < end) ? got : kd1;
                            fill = end;
                        }
                        if (kd0 != 0xffffffffu) add(kd0);
                        if (fill > 64u) { if (kd1 != 0xffffffffu) add(kd1); }
                    } else {
                        // (more than 128 records hold the row: slot by slot, as they sit)
#pragma unroll
                        for (int i = 0; i < RPL; ++i)
                            if ((hm[i] >> lane) & 1ull) add(per * lane + (uint32_t)i);
                    }
                    c4_wave_lds_sync();
                    int tot = 0;
                    if (16u * lane < NCOL) {
                        const uint4 lo4 = reinterpret_cast<const uint4 *>(s_wd)[2u * lane], hi4 = reinterpret_cast<const uint4 *>(s_wd)[2u * lane + 1u];
                        const uint32_t dws[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
                        for (int c = 15; c >= 0; --c) {
                            tot += (int)((dws[c >> 1] >> (16 * (c & 1))) & 0xffffu) - 0x4000;
                            wcol[c] = tot;                                  // columns c .. 15 of my 16
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 16; ++c) wcol[c] = 0;
                    }
                    const uint32_t incl = c4_wave_incl_add((uint32_t)tot);
                    const int right = (int)((uint32_t)__builtin_amdgcn_readlane((int)incl, 63) - incl);   // everything right of my 16 columns
#pragma unroll
                    for (int c = 0; c < 16; ++c) wcol[c] += right;
                } else {
#pragma unroll
                    for (int c = 0; c < 16; ++c) wcol[c] = 0;
                    const int col0 = (int)(16u * lane);                     // my first sample column
#pragma unroll
                    for (int i = 0; i < RPL; ++i) {
                        if (hm[i] == 0ull) continue;                        // (wave-uniform)
                        int J = 0; uint32_t step = 0u;
                        if ((hm[i] >> lane) & 1ull) evaluate(per * lane + (uint32_t)i, cy_r, J, step);
                        unsigned long long m = hm[i];
                        while (m) {                                         // every crossing to every lane: w(j) += step [j < J]
                            const int h = (int)__builtin_ctzll(m);
                            m &= m - 1ull;
                            const int n = __builtin_amdgcn_readlane(J, h) - col0;
                            const int sh = __builtin_amdgcn_readlane((int)step, h);
#pragma unroll
                            for (int c = 0; c < 16; ++c) wcol[c] += (c < n) ? sh : 0;
                        }
                    }
                }
                constexpr int PPL = 16 / NS;                              // pixels of my 16 sample columns
                int cq[PPL];
#pragma unroll
                for (int q = 0; q < PPL; ++q) {
                    cq[q] = 0;
#pragma unroll
                    for (int c = 0; c < NS; ++c) cq[q] += (wcol[NS * q + c] != 0);
                }
                int prev = __shfl_up(cq[PPL - 1], 1);
                if (lane == 0) prev = 0;
#pragma unroll
                for (int dq = 0; dq < PPL / 4; ++dq) {
                    const int before = dq ? cq[4 * dq - 1] : prev;
                    const uint32_t val = (uint32_t)(cq[4 * dq] - before) + ((uint32_t)(cq[4 * dq + 1] - cq[4 * dq]) << 8) +
                                         ((uint32_t)(cq[4 * dq + 2] - cq[4 * dq + 1]) << 16) + ((uint32_t)(cq[4 * dq + 3] - cq[4 * dq + 2]) << 24);
                    if (16u * lane < NCOL) atomicAdd(reinterpret_cast<uint32_t *>(s_E + (r >> LN) * L::EROW) + lane * (uint32_t)(PPL / 4) + (uint32_t)dq, val);
                }
                c4_wave_lds_sync();
            }
        }
        c4_wave_lds_sync();

        // ---- windows: lane = one 16-pixel window of one pixel row; integrate, map, one 16-byte store
        constexpr uint32_t K1 = 0x01010101u;
        // (one window pass: the hot form stores whole windows; a band or strip cut by the cell's border runs the same
        // arithmetic in a loop of its own — rolled, clipped stores — so that the hot loop stays as small as it was)
        // (The window addresses depend on the lane only, so the compiler computes them once per kernel, keeps them live across
        // the band loop and — in the 80-register instances — spills them; every reload is then a scratch load, and on gfx9 a
        // wave waits for a load with s_waitcnt vmcnt(0), which also waits for every pixel store it still has in flight: a band's
        // stores went out one HBM round trip at a time.  An opaque copy of the lane per band keeps the addresses where they
        // are used — a few integer instructions per window instead; and the store address is a 32-bit offset from the band's
        // wave-uniform base: PRB rows of < 2^27 bytes, fr_plan_render checks the pitch.)
        uint32_t lane_w = lane;
        asm volatile("" : "+v"(lane_w));
        const uint32_t wxw = lane_w & (NWIN - 1u);
        // (my window of pass 0, in E and in the output; pass `it` is 64 >> WLOG rows further down: a constant / a scalar away)
        const unsigned char *const e_lane = s_E + (lane_w >> WLOG) * L::EROW + 16u * wxw;
        const uint32_t out_lane = (lane_w >> WLOG) * (uint32_t)A.out_stride + 16u * wxw;
        auto window_pass = [&](uint32_t it, auto clipped) {
            const uint32_t prow = (lane_w >> WLOG) + it * (64u >> WLOG);
            const uint4 e = *reinterpret_cast<const uint4 *>(e_lane + it * (64u >> WLOG) * L::EROW);
            // inclusive byte prefix inside each dword: bytes 16 (i + 1) + sums; back to a bias of 16 per byte
            uint32_t x0 = e.x * K1, x1 = e.y * K1, x2 = e.z * K1, x3 = e.w * K1;
            x0 -= 0x30201000u;
            x1 = x1 + __builtin_amdgcn_perm(x0, x0, 0x03030303u) - 0x40302010u;
            x2 = x2 + __builtin_amdgcn_perm(x1, x1, 0x03030303u) - 0x40302010u;
            x3 = x3 + __builtin_amdgcn_perm(x2, x2, 0x03030303u) - 0x40302010u;
            // count entering my window = sum of the windows to my left in the pixel row
            const uint32_t T = (x3 >> 24) - 16u;                            // my window's total (signed)
            uint32_t inc = T;
            if (WLOG == 4) {
                inc += c4_dpp0<0x111>(inc);                                 // row_shr:1 within the 16 lanes of my pixel row
                inc += c4_dpp0<0x112>(inc);
                inc += c4_dpp0<0x114>(inc);
                inc += c4_dpp0<0x118>(inc);
            } else {
                // 8 (4) windows per pixel row: two (four) pixel rows share a DPP row — keep the scan inside each part
                uint32_t s;
                s = c4_dpp0<0x111>(inc); inc += (wxw >= 1u) ? s : 0u;
                s = c4_dpp0<0x112>(inc); inc += (wxw >= 2u) ? s : 0u;
                if (WLOG == 3) { s = c4_dpp0<0x114>(inc); inc += (wxw >= 4u) ? s : 0u; }
            }
            const uint32_t cin = inc - T;                                   // in [0, NS^2]
            const uint32_t cb4 = __builtin_amdgcn_perm(cin, cin, 0x00000000u);
            x0 += cb4; x1 += cb4; x2 += cb4; x3 += cb4;                     // bytes: 16 + k, k = inside samples of the pixel
            // u8 = round_half_up(255 k / NS^2) = 16 k - [k > 8] (NS = 4) / 64 k - [k > 2] (NS = 2), four pixels at once:
            // (16 + k) << s leaves 2^(4 + s) too much in every byte — 0x01010100 (0x04040400) over the dword, mod 2^32
            auto map4 = [](uint32_t x) -> uint32_t {
                uint32_t r;
                if (NS == 4) {
                    const uint32_t t = ((x + 0x07070707u) >> 5) & 0x01010101u;
                    const uint32_t u = 0xfefeff00u - t;
                    asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(r) : "v"(x), "v"(u));      // (x << 4) + u in one instruction
                } else {
                    const uint32_t t = ((x + 0x0d0d0d0du) >> 5) & 0x01010101u;
                    const uint32_t u = 0xfbfbfc00u - t;
                    asm("v_lshl_add_u32 %0, %1, 6, %2" : "=v"(r) : "v"(x), "v"(u));
                }
                return r;
            };
            const uint4 v = make_uint4(map4(x0), map4(x1), map4(x2), map4(x3));
            unsigned char *dst = (out_band + (size_t)(it * (64u >> WLOG)) * A.out_stride) + out_lane;
            if (!decltype(clipped)::value) c4_store16(dst, v);
            else c4_store_clip(dst, v, prow < hlim ? (int)wlim - (int)(16u * wxw) : 0);
        };
        if (C4_ABL_KEEP(1)) {
            if (__builtin_expect(!edge, 1)) {
#pragma unroll
                for (uint32_t it = 0; it < (PRB * NWIN) / 64u; ++it) window_pass(it, std::false_type{});
            } else {
#pragma clang loop unroll(disable)
                for (uint32_t it = 0; it < (PRB * NWIN) / 64u; ++it) window_pass(it, std::true_type{});
            }
        }
        c4_wave_lds_sync();                        // E is the next band's list region
    }
}

#ifdef FR_C4_STATS
extern "C" int fr_debug_read_c4_stats(unsigned long long *out16, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_c4_stats), sizeof(g_c4_stats));
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_c4_stats), z, sizeof z);
    }
    return e == hipSuccess ? 0 : -2;
}
#endif
uint32_t cov4_wg_waves() { return C4_WAVES; }
uint32_t cov4_max_segments() { return 768u; }     // (with 1024 record slots; 384 with 512, 256 for the smaller instances: fr_plan_create)

template <int WLOG, int RPL, int NS>
static hipError_t cov4_launch_cap(const RenderArgs &a, dim3 grid, hipStream_t stream, char *name, size_t name_cap)
{
    // glyphs of <= 128 candidate roots (RPL == 2) all but never put more than 16 crossings on a sample row (a real font:
    // 1 row in 100 000): their instance keeps 16 per row in registers — half the list to initialise, pull and sort, 5 %
    // faster — and the rare fuller row takes the direct sum like any over-full row
    const uint32_t kmax = (RPL == 2 && a.kmax > 16u) ? 16u : a.kmax;
    // (the 1024-record instance — glyphs of 385 .. 768 segments, rare — exists with 32 kept crossings only)
    const int cap = RPL >= 16 ? 32 : (kmax <= 8 ? 8 : (kmax <= 16 ? 16 : 32));
    // the instance as rocprofv3 names it
    if (name) snprintf(name, name_cap, "fr::cov4_kernel<%d, %d, %d, %d>", WLOG, cap, RPL, NS);
    if (!grid.x) return hipSuccess;                // (name only)
    const size_t lds = (cap == 8 ? C4Lds<WLOG, RPL, NS, 8>::TOTAL : (cap == 16 ? C4Lds<WLOG, RPL, NS, 16>::TOTAL : C4Lds<WLOG, RPL, NS, 32>::TOTAL)) + a.lds_pad;
    auto launch = [&](auto kern) -> hipError_t {
        if (lds > 48 * 1024) {                   // (the 512-record instance; below that the default limit is enough)
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(64 * C4_WAVES), lds, stream, a);
        return hipGetLastError();
    };
    if constexpr (RPL < 16) {
        if (cap == 8) return launch(cov4_kernel<WLOG, 8, RPL, NS>);
        if (cap == 16) return launch(cov4_kernel<WLOG, 16, RPL, NS>);
    }
    return launch(cov4_kernel<WLOG, 32, RPL, NS>);
}

template <int WLOG, int NS>
static hipError_t cov4_launch_rpl(const RenderArgs &a, uint32_t rec_cap, dim3 grid, hipStream_t stream, char *name, size_t name_cap)
{
    if (rec_cap <= 128u) return cov4_launch_cap<WLOG, 2, NS>(a, grid, stream, name, name_cap);
    if (rec_cap <= 256u) return cov4_launch_cap<WLOG, 4, NS>(a, grid, stream, name, name_cap);
    if (rec_cap > 512u) return cov4_launch_cap<WLOG, 16, NS>(a, grid, stream, name, name_cap);
    return cov4_launch_cap<WLOG, 8, NS>(a, grid, stream, name, name_cap);
}

// jobs: cells of any size up to 2048 / ns sample rows (strips of a.strip_w in {64, 128, 256} pixels, wave bands of 64 / ns
// pixel rows; the last of each may be partial), ns x ns samples (ns in {2, 4}), every glyph with <= 384 segments and
// <= rec_cap (128, 256 or 512) possible root records (checked by fr_plan_create).  launch = false: only name the
// instance (as rocprofv3 prints it) into `name`.
hipError_t launch_cov4(const RenderArgs &a, uint32_t rec_cap, int ns, hipStream_t stream, bool launch, char *name, size_t name_cap)
{
    const dim3 grid(launch ? (uint32_t)((size_t)a.n_jobs * a.band_groups * a.strips) : 0u);
    if (ns == 4) {
        if (a.strip_w == 256u) return cov4_launch_rpl<4, 4>(a, rec_cap, grid, stream, name, name_cap);
        if (a.strip_w == 128u) return cov4_launch_rpl<3, 4>(a, rec_cap, grid, stream, name, name_cap);
        if (a.strip_w == 64u) return cov4_launch_rpl<2, 4>(a, rec_cap, grid, stream, name, name_cap);
    } else if (ns == 2) {
        if (a.strip_w == 256u) return cov4_launch_rpl<4, 2>(a, rec_cap, grid, stream, name, name_cap);
        if (a.strip_w == 128u) return cov4_launch_rpl<3, 2>(a, rec_cap, grid, stream, name, name_cap);
        if (a.strip_w == 64u) return cov4_launch_rpl<2, 2>(a, rec_cap, grid, stream, name, name_cap);
    }
    return hipErrorInvalidValue;
}

}  // namespace fr
