// fr_c4.hpp — what cov4_kernel (fr_cov4.hip, 16 samples per pixel) and win1_kernel (fr_win1.hip, 1 sample per pixel)
// share: the 40-byte root record, wave64 DPP scans, the workgroup set-up (records with exact row ranges + cx table).
#pragma once
#include "fr_records.hpp"

namespace fr {

#ifndef FR_C4_WAVES
#define FR_C4_WAVES 4
#endif
#ifndef FR_C4_OCC
#define FR_C4_OCC 4
#endif
#ifndef FR_C4_PCAP
#define FR_C4_PCAP 448
#endif
#ifndef FR_C4_OCC_SMALL
#define FR_C4_OCC_SMALL 6
#endif
#ifndef FR_DYN_BANDS
#define FR_DYN_BANDS 1
#endif
#ifndef FR_W1_OCC_SMALL
#define FR_W1_OCC_SMALL 6
#endif
// PCAP: (record, row) pairs laid out per round (a multiple of 64).
enum { C4_WAVES = FR_C4_WAVES, C4_PCAP = FR_C4_PCAP };
// u16 slots per row list: the CAP crossings a row keeps + the dump slot + padding to a multiple of 4 (8-byte rows).
// (36 for CAP = 32; the instances that keep 16 or 8 have shorter rows and leave the LDS to more workgroups)
constexpr uint32_t c4_lstride(int cap) { return (uint32_t)cap + 4u; }
// Waves per SIMD the register allocation is held to.  The instances that keep <= 16 crossings per row with two records
// per lane (glyphs of <= 128 candidate roots: ASCII-like, a real font) need <= 26 KB of LDS per workgroup, so SIX
// workgroups fit a CU if the kernel stays within 80 VGPRs: measured 0.0754 -> 0.0643 ms on configs[3]'s shard (128^2
// cells), 0.343 -> 0.291 ms on 256^2 cells of 32 segments — and nothing at five (an odd number of waves per SIMD: DESIGN.md
// section 4.0).  The instances that keep 32 crossings or more records are bound to four (three) by their LDS.
// (the 1024-record instances: 77 KB of LDS, two workgroups per CU — the register budget of two waves per SIMD)
// (the 512-record instances: 45 - 53 KB of LDS, three workgroups per CU — and a third of the register file each, which the
// second register list of their two-pass bands needs: fr_cov4.hip, SPLIT)
constexpr int c4_occ(int cap, int wlog, int rpl) { return rpl >= 16 ? 2 : (rpl >= 8 ? 3 : ((cap <= 16 && rpl == 2) ? FR_C4_OCC_SMALL : FR_C4_OCC)); }
constexpr int w1_occ(int rpl, int wlog) { return rpl >= 16 ? 2 : ((rpl == 2) ? FR_W1_OCC_SMALL : FR_C4_OCC); }
// 16 bytes of a row list (8-byte aligned): two 8-byte LDS accesses unless the rows are 16-byte aligned
template <uint32_t LSTRIDE>
__device__ __forceinline__ uint4 c4_ld16(const uint16_t *p)
{
    if ((LSTRIDE * 2) % 16 == 0) return *reinterpret_cast<const uint4 *>(p);
    const uint2 a = reinterpret_cast<const uint2 *>(p)[0], b = reinterpret_cast<const uint2 *>(p)[1];
    return make_uint4(a.x, a.y, b.x, b.y);
}
template <uint32_t LSTRIDE>
__device__ __forceinline__ void c4_st16(uint16_t *p, uint4 v)
{
    if ((LSTRIDE * 2) % 16 == 0) { *reinterpret_cast<uint4 *>(p) = v; return; }
    reinterpret_cast<uint2 *>(p)[0] = make_uint2(v.x, v.y);
    reinterpret_cast<uint2 *>(p)[1] = make_uint2(v.z, v.w);
}

// 40-byte root record of the cov4 kernel (LDS only)
struct __attribute__((aligned(8))) Rec40 {
    float a;        // quadratic: p0y - 2 p1y + p2y (:48); linear (a == 0 branch): the divisor p2y - p0y (:51)
    float b;        // quadratic: p0y - p1y;               linear: p0y
    float c1, c2;   // quadratic: p1y^2, p0y p2y (:58);    linear: 0
    float ax, bx, p0x;   // (:53 / :65)
    float rden;     // RN(1 / a)
    float sgn;      // +1 for the t+ root, -1 for t-  (sqrt * +-1 is exact: B - sqrt == B + (-sqrt))
    uint32_t fr;    // ra | re << 12 | cb << 24 | zb << 26 | linear << 31 (ra, re <= 2048): sample rows [ra, re) accept the root;
                    // the crossing's step code is (dy > 0) ? zb : cb   (2: +1, 0: -1; :55, :68)
};

template <int CTRL>
__device__ __forceinline__ uint32_t c4_dpp0(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t c4_dppm(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t c4_wave_incl_add(uint32_t x)
{
    x += c4_dppm<0x111, 0xf>(x);
    x += c4_dppm<0x112, 0xf>(x);
    x += c4_dppm<0x114, 0xf>(x);
    x += c4_dppm<0x118, 0xf>(x);
    x += c4_dppm<0x142, 0xa>(x);
    x += c4_dppm<0x143, 0xc>(x);
    return x;
}
__device__ __forceinline__ uint32_t c4_wave_incl_max(uint32_t x)
{
    x = max(x, c4_dppm<0x111, 0xf>(x));
    x = max(x, c4_dppm<0x112, 0xf>(x));
    x = max(x, c4_dppm<0x114, 0xf>(x));
    x = max(x, c4_dppm<0x118, 0xf>(x));
    x = max(x, c4_dppm<0x142, 0xa>(x));
    x = max(x, c4_dppm<0x143, 0xc>(x));
    return x;
}
// select on a wave mask held in SGPRs — the VOP3 form, whose cost does not depend on what wrote the mask
// (a VOP2 v_cndmask reading a VCC that is not fresh costs 3-5 vector instructions: profiles/r02/issue_model3.txt)
__device__ __forceinline__ uint32_t c4_sel(unsigned long long m, uint32_t if_set, uint32_t if_clear)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}
__device__ __forceinline__ float c4_self(unsigned long long m, float if_set, float if_clear)
{
    return __builtin_bit_cast(float, c4_sel(m, __builtin_bit_cast(uint32_t, if_set), __builtin_bit_cast(uint32_t, if_clear)));
}
__device__ __forceinline__ void c4_wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}


// 16-byte streaming store of finished pixels (never read again by this kernel): the non-temporal hint keeps them from
// displacing the tables the kernel does re-read in L2
__device__ __forceinline__ void c4_store16(void *dst, uint4 v)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 u32x4_u __attribute__((aligned(4)));
    u32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4_u *>(dst));
}

// a settled candidate as the kernels keep it (40 bytes): row range [ra, re) in the low bits of `fr`
__device__ __forceinline__ Rec40 c4_make_rec40(const Rec &r, uint32_t ra, uint32_t re)
{
    Rec40 m;
    const bool lin = (int32_t)r.flags < 0;
    m.a = lin ? r.c1 : r.a;
    m.b = r.b; m.c1 = lin ? 0.0f : r.c1; m.c2 = r.c2;
    m.ax = r.ax; m.bx = r.bx; m.p0x = r.p0x; m.rden = r.rden;
    m.sgn = r.sqsign ? -1.0f : 1.0f;
    const uint32_t cb = lin ? (r.flags & REC_LIN_PLUS) : 2u, zb = lin ? cb : 0u;
    m.fr = ra | (re << 12) | (cb << 24) | (zb << 26) | (lin ? 0x80000000u : 0u);
    return m;
}

// Set-up for glyphs of <= 32 segments (<= 64 candidate roots: one per lane of ONE wave — ASCII-like glyphs), four
// waves.  Settling a candidate's row range costs four evaluations of the reference's acceptance at the rows around
// the two guessed ends; with the plain set-up wave 0 would do all of it while waves 1 - 3 (and the SIMDs they sit
// on) wait.  Here every wave prepares the same 64 candidates and looks at ONE of the four rows (wave 0: ra - 1,
// 1: ra, 2: re - 1, 3: re); wave 0 takes the guess when the four classes confirm it — exactly the condition under
// which record_settle's walks would not move — and walks as before otherwise.  Same records, same order.
template <uint32_t RCAP, int N, uint32_t NCOL>
__device__ __forceinline__ uint32_t c4_setup_small(const RenderArgs &A, const Job &job, uint32_t seg0, uint32_t nseg, uint32_t x0s,
                                                   int phase, float *s_cxp, Rec40 *s_rec, uint32_t *s_wcnt, uint32_t *s_tmp, const float *cyt = nullptr)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t Hs = job.h * (uint32_t)N;
    RowGeom geo;
    geo.max_y = job.max_y; geo.scale = job.scale; geo.rows = Hs; geo.n = N; geo.phase = phase; geo.cyt = cyt;
    const bool have = lane < 2u * nseg;
    Rec r;
    RowGuess g;
    g.empty = true; g.ra = 1u; g.re = 0u;
    uint32_t cls = 3u;                                  // 3: no such row / not asked — the condition it stands for holds
    if (have) {
        record_prep(A.seg_pts + 6u * (size_t)(seg0 + (lane >> 1)), lane & 1u, geo, r, g);
        if (!g.empty) {
            const uint32_t row = (wave == 0u) ? g.ra - 1u : (wave == 1u) ? g.ra : (wave == 2u) ? g.re - 1u : g.re;   // (ra - 1 wraps past the cell at ra = 0)
            if (row < Hs && (wave != 2u || g.re > g.ra)) cls = (uint32_t)classify_row(r, geo.cy(row));
        }
    }
    s_tmp[tid] = cls;
    // exact sample abscissae of this strip: cx(j) = (f32(min_x + x) + off(i)) / scale   (:26)
    const int32_t min_xs = job.min_x + (int32_t)x0s;
    for (uint32_t j = tid; j < NCOL; j += 256u)
        s_cxp[1u + j] = ((float)(min_xs + (int32_t)(j / (uint32_t)N)) + sub_off((int)(j % (uint32_t)N), N, phase)) / job.scale;
    if (tid == 2) s_cxp[0] = -__builtin_inff();
    if (tid == 3) s_cxp[1u + NCOL] = __builtin_inff();
    __syncthreads();
    if (wave == 0u) {
        Rec40 mine;
        mine.fr = 0u;
        bool live = false;
        if (have) {
            uint32_t ra = g.ra, re = g.re;
            if (!g.empty) {
                const uint32_t c0 = s_tmp[lane], c1 = s_tmp[64u + lane], c2 = s_tmp[128u + lane], c3 = s_tmp[192u + lane];
                // record_settle would leave (ra, re) alone iff: the row above ra is rejected from above (class 2), ra is
                // not (class <= 1), re >= ra, the row before re is accepted (class >= 1), re is below the set (class 0)
                const bool ok = (c0 >= 2u) && (c1 <= 1u || c1 == 3u) && re >= ra && (c2 >= 1u) && (c3 == 0u || c3 == 3u);
                if (!ok) record_settle(r, geo, ra, re);
            }
            live = ra < re;
            mine = c4_make_rec40(r, ra, re);
        }
        // quadratic records first, linear ones last (as c4_setup)
        const bool linr = live && (int32_t)mine.fr < 0;
        const unsigned long long lm = __ballot(live && !linr), ll = __ballot(linr), below = (1ull << lane) - 1ull;
        const uint32_t n_quad = (uint32_t)__popcll(lm);
        uint32_t pos = RCAP;
        if ((lm >> lane) & 1ull) pos = (uint32_t)__popcll(lm & below);
        if ((ll >> lane) & 1ull) pos = n_quad + (uint32_t)__popcll(ll & below);
        if (pos < RCAP) s_rec[pos] = mine;
        if (lane == 0u) s_wcnt[0] = min(n_quad + (uint32_t)__popcll(ll), RCAP);
    }
    __syncthreads();
    return s_wcnt[0];
}

// The same for 33 .. 64 segments (<= 128 candidate roots: one per lane of waves 0 and 1).  Thread c + 128 h looks at the
// rows around the upper end (h = 0: ra - 1 and ra) or the lower end (h = 1: re - 1 and re) of candidate c; the owners
// (h = 0) take the guess when all four classes confirm it and walk otherwise.
template <uint32_t RCAP, int N, uint32_t NCOL>
__device__ __forceinline__ uint32_t c4_setup_mid(const RenderArgs &A, const Job &job, uint32_t seg0, uint32_t nseg, uint32_t x0s,
                                                 int phase, float *s_cxp, Rec40 *s_rec, uint32_t *s_wcnt, uint32_t *s_tmp, const float *cyt = nullptr)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t c = tid & 127u, h = tid >> 7;
    const uint32_t Hs = job.h * (uint32_t)N;
    RowGeom geo;
    geo.max_y = job.max_y; geo.scale = job.scale; geo.rows = Hs; geo.n = N; geo.phase = phase; geo.cyt = cyt;
    const bool have = c < 2u * nseg;
    Rec r;
    RowGuess g;
    g.empty = true; g.ra = 1u; g.re = 0u;
    uint32_t cls0 = 3u, cls1 = 3u;                      // 3: no such row / not asked
    if (have) {
        record_prep(A.seg_pts + 6u * (size_t)(seg0 + (c >> 1)), c & 1u, geo, r, g);
        if (!g.empty) {
            const uint32_t end = h ? g.re : g.ra;       // rows end - 1 and end
            if (end - 1u < Hs && (h == 0u || g.re > g.ra)) cls0 = (uint32_t)classify_row(r, geo.cy(end - 1u));
            if (end < Hs) cls1 = (uint32_t)classify_row(r, geo.cy(end));
        }
    }
    s_tmp[tid] = cls0 | (cls1 << 8);
    const int32_t min_xs = job.min_x + (int32_t)x0s;
    for (uint32_t j = tid; j < NCOL; j += 256u)
        s_cxp[1u + j] = ((float)(min_xs + (int32_t)(j / (uint32_t)N)) + sub_off((int)(j % (uint32_t)N), N, phase)) / job.scale;
    if (tid == 2) s_cxp[0] = -__builtin_inff();
    if (tid == 3) s_cxp[1u + NCOL] = __builtin_inff();
    __syncthreads();
    Rec40 mine;
    mine.fr = 0u;
    bool live = false;
    if (h == 0u && have) {
        uint32_t ra = g.ra, re = g.re;
        if (!g.empty) {
            const uint32_t up = s_tmp[c], dn = s_tmp[128u + c];
            const uint32_t c0 = up & 0xffu, c1 = up >> 8, c2 = dn & 0xffu, c3 = dn >> 8;
            const bool ok = (c0 >= 2u) && (c1 <= 1u || c1 == 3u) && re >= ra && (c2 >= 1u) && (c3 == 0u || c3 == 3u);
            if (!ok) record_settle(r, geo, ra, re);
        }
        live = ra < re;
        mine = c4_make_rec40(r, ra, re);
    }
    // quadratic records first, linear ones last (as c4_setup): waves 0 and 1 hold them
    const bool linr = live && (int32_t)mine.fr < 0;
    const unsigned long long lm = __ballot(live && !linr), ll = __ballot(linr), below = (1ull << lane) - 1ull;
    if (lane == 0u && wave < 2u) { s_wcnt[wave] = (uint32_t)__popcll(lm); s_wcnt[2u + wave] = (uint32_t)__popcll(ll); }
    __syncthreads();
    const uint32_t q0 = s_wcnt[0], q1 = s_wcnt[1], l0 = s_wcnt[2], l1 = s_wcnt[3];
    const uint32_t n_quad = q0 + q1;
    uint32_t pos = RCAP;
    if ((lm >> lane) & 1ull) pos = (wave ? q0 : 0u) + (uint32_t)__popcll(lm & below);
    if ((ll >> lane) & 1ull) pos = n_quad + (wave ? l0 : 0u) + (uint32_t)__popcll(ll & below);
    if (wave < 2u && pos < RCAP) s_rec[pos] = mine;
    __syncthreads();
    return min(n_quad + l0 + l1, RCAP);
}

// Set-up of a workgroup of NW waves for one cell strip: candidate roots tid, tid + 64 NW, ... (<= 256 segments; <= 384 where RCAP = 512) with the
// exact range of this cell's sample rows that accept each (fr_records.hpp), compacted into s_rec (<= RCAP kept), and
// the padded table of the strip's exact sample abscissae.  Two workgroup barriers.  -> number of records.
// N: samples per pixel axis (4: cov4_kernel, 1: win1_kernel); NCOL: sample columns of the strip.
template <uint32_t NW, uint32_t RCAP, int N, uint32_t NCOL>
__device__ __forceinline__ uint32_t c4_setup(const RenderArgs &A, const Job &job, uint32_t seg0, uint32_t nseg, uint32_t x0s,
                                             int phase, float *s_cxp, Rec40 *s_rec, uint32_t *s_wcnt, uint32_t *s_tmp, const float *cyt = nullptr)
{
    if (NW == 4u && 2u * nseg <= 64u)                                        // (workgroup-uniform; s_tmp: 1 KB, free until the bands start)
        return c4_setup_small<RCAP, N, NCOL>(A, job, seg0, nseg, x0s, phase, s_cxp, s_rec, s_wcnt, s_tmp, cyt);
    if (NW == 4u && 2u * nseg <= 128u)
        return c4_setup_mid<RCAP, N, NCOL>(A, job, seg0, nseg, x0s, phase, s_cxp, s_rec, s_wcnt, s_tmp, cyt);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t Hs = job.h * (uint32_t)N;
    // candidates per thread: <= 768 segments where 1024 records are kept, <= 384 where 512, else <= 256
    constexpr uint32_t CPT = (RCAP >= 1024u ? 1536u : (RCAP >= 512u ? 768u : 512u)) / (64u * NW);
    s_wcnt = s_tmp;                                                          // (2 CPT NW counters: the scratch block has the room)
    Rec40 mine[CPT];
    unsigned long long lm[CPT], ll[CPT];
#pragma unroll
    for (uint32_t it = 0; it < CPT; ++it) {
        const uint32_t c = tid + it * 64u * NW;
        bool live = false;
        lm[it] = 0ull; ll[it] = 0ull;
        mine[it].fr = 0u;
        if (it > 0 && it * 64u * NW >= 2u * nseg) continue;                 // workgroup-uniform: no second candidate
        if (c < 2u * nseg) {
            Rec r;
            RowGeom geo;
            geo.max_y = job.max_y; geo.scale = job.scale; geo.rows = Hs; geo.n = N; geo.phase = phase; geo.cyt = cyt;
            build_record_rows(A.seg_pts + 6u * (size_t)(seg0 + (c >> 1)), c & 1u, geo, r);
            const uint32_t ra = __builtin_bit_cast(uint32_t, r.lo), re = __builtin_bit_cast(uint32_t, r.hi);
            live = ra < re;
            mine[it] = c4_make_rec40(r, ra, re);
        }
        // quadratic records first, linear ones (the a == 0 branch) last: the pair sequence follows the record order,
        // so an evaluation trip is almost always all-quadratic or all-linear and takes a body without the other's work
        const bool linr = live && (int32_t)mine[it].fr < 0;
        lm[it] = __ballot(live && !linr);
        ll[it] = __ballot(linr);
        if (lane == 0) { s_wcnt[it * NW + wave] = (uint32_t)__popcll(lm[it]); s_wcnt[CPT * NW + it * NW + wave] = (uint32_t)__popcll(ll[it]); }
    }
    // exact sample abscissae of this strip: cx(j) = (f32(min_x + x) + off(i)) / scale   (:26)
    const int32_t min_xs = job.min_x + (int32_t)x0s;
    for (uint32_t j = tid; j < NCOL; j += 64u * NW)
        s_cxp[1u + j] = ((float)(min_xs + (int32_t)(j / (uint32_t)N)) + sub_off((int)(j % (uint32_t)N), N, phase)) / job.scale;
    if (tid == 2) s_cxp[0] = -__builtin_inff();
    if (tid == 3) s_cxp[1u + NCOL] = __builtin_inff();
    __syncthreads();
    uint32_t rec_cnt = 0;
    {
        uint32_t base_q[CPT], base_l[CPT], n_quad = 0;
#pragma unroll
        for (uint32_t it = 0; it < CPT; ++it) base_q[it] = base_l[it] = 0;
#pragma unroll
        for (uint32_t q = 0; q < CPT * NW; ++q) {
            if (q >= NW && q / NW * 64u * NW >= 2u * nseg) continue;        // (never written)
            const uint32_t cq = s_wcnt[q], cl = s_wcnt[CPT * NW + q];
#pragma unroll
            for (uint32_t it = 0; it < CPT; ++it) {
                base_q[it] += (q < it * NW + wave) ? cq : 0u;
                base_l[it] += (q < it * NW + wave) ? cl : 0u;
            }
            n_quad += cq;
            rec_cnt += cq + cl;
        }
#pragma unroll
        for (uint32_t it = 0; it < CPT; ++it) {
            const unsigned long long below = (1ull << lane) - 1ull;
            uint32_t pos = RCAP;
            if ((lm[it] >> lane) & 1ull) pos = base_q[it] + (uint32_t)__popcll(lm[it] & below);
            if ((ll[it] >> lane) & 1ull) pos = n_quad + base_l[it] + (uint32_t)__popcll(ll[it] & below);
            if (pos < RCAP) s_rec[pos] = mine[it];                          // (the plan only sends glyphs with <= RCAP possible records)
        }
    }
    rec_cnt = min(rec_cnt, RCAP);
    __syncthreads();
    return rec_cnt;
}

}  // namespace fr
