// fr_sdf.hip — per-pixel signed distance field (BASELINE configs[4]).  BUILD-DEFINED: the
// reference has no SDF; the definition below is this library's.  The test checker
// keeps a CPU twin written from this definition (same operations in the same order, no FMA
// contraction, IEEE divide/sqrt), so the two agree bit for bit.
//
//   sample     : the pixel's sample point exactly as renderGlyph's (render_glyph.zig:26-27),
//                n = 1, phase corner or centre
//   distance   : min over the glyph's quadratic segments B(t) = p0 + 2t(p1-p0) + t^2(p0-2p1+p2),
//                t in [0,1], of |B(t) - q| (font units): 9 uniform probes, then 4 Newton steps on
//                (B(t)-q).B'(t) = 0 clamped to the probe's neighbourhood; times `scale` -> pixels
//   sign       : + inside (winding != 0, the reference's non-zero test on its own winding number,
//                render_glyph.zig:29,35-73), - outside; one BIT per pixel in a job-local bit plane that
//                win1_kernel's sign-bit mode wrote right before this kernel (an eighth of the bytes of a mask, and
//                every output byte is then written exactly once, by this kernel) — or, for the few jobs only the
//                general render_kernel takes, the byte (255 / 0) its 1-sample coverage left in the output
//   encoding   : u8 = clamp(floor(128 + 16*d + 0.5), 0, 255)   (8 pixels of range either side)
//
// Shape: one WAVE (a 64-lane workgroup, no barriers) per 64 x 16-pixel region of a cell, taken as 16 quads of
// 8x8 pixels, one lane per pixel.  The encoding saturates 8 pixels from the outline, so a segment whose
// bounding box is farther than that from a set of sample points cannot change any of their bytes.  The box
// is ORIENTED along the chord: B(t) = chord(t) + 2t(1-t) v with v = p1 - (p0 + p2)/2 and 2t(1-t) in [0, 1/2],
// so in the frame (tau, n) of the chord the curve stays inside [min(0, va/2), L + max(0, va/2)] x
// [min(0, vb/2), max(0, vb/2)] — for a long diagonal or gently bent segment a far thinner set than the
// axis-aligned box of its control points.  It is used three times, each time exactly:
//   region : the wave reads the glyph's segments 64 at a time (one per lane, 12 bytes) and keeps those within
//            reach of the region's sample box — ballot + lane-prefix compaction into an LDS list that also
//            holds what does not depend on the sample: A, the second difference and the probe points B(k/8);
//   quad   : the list's entries against the 16 quads' sample boxes (4 entries x 16 quads per step, each box projected
//            onto the segment's frame); the ballots are the quads' candidate sets.
//            A quad without candidates is "farther than 8 pixels": its bytes are the saturated encoding, 255 / 0
//            by the sign bit (a whole region without candidates: one 16-byte store per lane);
//   pixel  : per candidate, the pixel's own box distance against the reach and against the best so far.
// The distance the kernel computes is the distance to SOME point of the curve, hence >= the box distance: every
// skipped segment would have produced a value past the clamp or not below the minimum, so the minimum over the
// segments that are evaluated is the minimum over all of them, whatever the order (option "sdf_cull" = 0 skips
// nothing; tests compare the two).  Glyphs of more than 64 segments: the per-pixel minima of a region wait in
// LDS between blocks.
#include "fr_device.hpp"

namespace fr {

// a wave's region: 64 x 16 pixels = 8 x 2 quads of 8 x 8 pixels.  Its rows are 64 bytes: the finished region leaves as
// whole 64-byte runs — the unit the memory side moves — where the 32-byte rows of a 32 x 32 region cost a read of the other
// half of every run (profiles/r03: 2.8 x the algorithmic bytes moved with 32 x 32 regions)
constexpr uint32_t SDF_RW = 64u, SDF_RH = 16u, SDF_QX = SDF_RW / 8u;
constexpr uint32_t SDF_BLOCK = 64u;           // segments read per block, one per lane (48, for 16 waves per CU instead of 13, was no faster: the kernel is VALU-bound)
constexpr uint32_t SDF_ENTRY = 28u;           // floats per staged segment (7 x 16 bytes):
// [0..3] p0x p0y Ax Ay   [4..7] tau_x tau_y a_lo a_hi   [8..11] b_lo b_hi Dx Dy (second difference)
// [12..19] Bx(k/8), k = 1..8   [20..27] By(k/8)

// |B(t) - q|^2 minimised over t in [0,1] as the header defines it; `e` is the staged entry.  The probe points
// B(k/8) = (p0 + (2t) A) + (t t) D are the same roundings as the header's expression tree — 2t and t t are exact
// for t = k/8 — only hoisted out of the pixel.
// `want`: this lane's result will be used (the others ride along and must not hold the wave in the Newton loop)
__device__ __forceinline__ float seg_dist2(const float *e, float qx, float qy, bool want = true)
{
    const float4 pa = *reinterpret_cast<const float4 *>(e);
    const float2 dd = *reinterpret_cast<const float2 *>(e + 10);
    const float p0x = pa.x, p0y = pa.y, ax = pa.z, ay = pa.w, bx = dd.x, by = dd.y;
    float best = 3.402823466e+38f, bt = 0.0f;
    {
        const float x = p0x - qx, y = p0y - qy;                          // k = 0: B(0) = p0
        const float d2 = x * x + y * y;
        if (d2 < best) { best = d2; bt = 0.0f; }
    }
    float px[8], py[8];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float4 vx = *reinterpret_cast<const float4 *>(e + 12 + 4 * k), vy = *reinterpret_cast<const float4 *>(e + 20 + 4 * k);
        px[4 * k] = vx.x; px[4 * k + 1] = vx.y; px[4 * k + 2] = vx.z; px[4 * k + 3] = vx.w;
        py[4 * k] = vy.x; py[4 * k + 1] = vy.y; py[4 * k + 2] = vy.z; py[4 * k + 3] = vy.w;
    }
#pragma unroll
    for (int k = 1; k <= 8; ++k) {
        const float x = px[k - 1] - qx, y = py[k - 1] - qy;
        const float d2 = x * x + y * y;
        if (d2 < best) { best = d2; bt = (float)k * 0.125f; }
    }
    const float lo = fmaxf(bt - 0.125f, 0.0f), hi = fminf(bt + 0.125f, 1.0f);
    float t = bt;
    for (int it = 0; it < 4; ++it) {
        const float x = p0x + 2.0f * t * ax + t * t * bx - qx;
        const float y = p0y + 2.0f * t * ay + t * t * by - qy;
        const float dx = 2.0f * (ax + t * bx), dy = 2.0f * (ay + t * by);
        const float f = x * dx + y * dy;
        const float fp = dx * dx + dy * dy + 2.0f * (x * bx + y * by);
        const float t0 = t;
        if (fp > 0.0f) t = t - f / fp;
        t = fminf(fmaxf(t, lo), hi);
        // a step is a function of t alone: once NO lane's t moved, the remaining steps would return the same t again —
        // leaving here gives the very bytes of four steps (the CPU twin always takes four)
        if (__all(!want || t == t0)) break;
    }
    {
        const float x = p0x + 2.0f * t * ax + t * t * bx - qx;
        const float y = p0y + 2.0f * t * ay + t * t * by - qy;
        const float d2 = x * x + y * y;
        if (d2 < best) best = d2;
    }
    return best;
}

// The oriented box of a segment: unit chord direction tau (n = (-tau_y, tau_x)), the box [a_lo, a_hi] x [b_lo, b_hi] in
// that frame about p0.  A degenerate chord (p0 == p2) takes the x axis.  The frame is only as good as float: |tau|
// is 1 within a few 2^-24 and the box edges carry a few roundings of magnitudes <= 2^17 — all far inside the
// `slack` every user of obb_gap2 subtracts.
struct Obb { float tx, ty, alo, ahi, blo, bhi; };

__device__ __forceinline__ Obb make_obb(const float *c)
{
    const float ux = c[4] - c[0], uy = c[5] - c[1];
    const float len = __builtin_sqrtf(ux * ux + uy * uy);
    Obb o;
    o.tx = len > 0.0f ? ux / len : 1.0f;
    o.ty = len > 0.0f ? uy / len : 0.0f;
    const float vx = c[2] - 0.5f * (c[0] + c[4]), vy = c[3] - 0.5f * (c[1] + c[5]);     // v = p1 - midpoint (exact for i16 points)
    const float va = 0.5f * (vx * o.tx + vy * o.ty), vb = 0.5f * (vy * o.tx - vx * o.ty);
    o.alo = fminf(va, 0.0f); o.ahi = len + fmaxf(va, 0.0f);
    o.blo = fminf(vb, 0.0f); o.bhi = fmaxf(vb, 0.0f);
    return o;
}

// squared distance from q to the oriented box after the box has been grown by `grow` on every side (0 inside)
__device__ __forceinline__ float obb_gap2(float p0x, float p0y, float tx, float ty, float alo, float ahi, float blo, float bhi,
                                          float qx, float qy, float grow)
{
    const float wx = qx - p0x, wy = qy - p0y;
    const float a = wx * tx + wy * ty, b = wy * tx - wx * ty;
    const float da = fmaxf(fmaxf(alo - a, a - ahi) - grow, 0.0f), db = fmaxf(fmaxf(blo - b, b - bhi) - grow, 0.0f);
    return da * da + db * db;
}

// MULTI: some glyph of the batch has more than SDF_BLOCK segments (the minima then wait in 4 KB of LDS between blocks;
// without them 21 waves fit a CU instead of 13)
#ifdef FR_SDF_STATS
__device__ unsigned long long g_sdf_stats[4];    // diagnostic build: distance evaluations (wave level), lanes that wanted them
extern "C" int fr_debug_read_sdf_stats(unsigned long long *out4)
{
    return hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_sdf_stats), sizeof(g_sdf_stats)) == hipSuccess ? 0 : -2;
}
#endif
template <bool MULTI>
__global__ __launch_bounds__(64) void sdf_kernel(const Job *__restrict__ jobs, const uint32_t *__restrict__ job_seg,
                                                 const int16_t *__restrict__ seg_pts, uint8_t *__restrict__ out,
                                                 uint64_t out_stride, uint32_t regions_x, uint32_t regions_y,
                                                 int phase_center, int cull, const uint32_t *__restrict__ bits,
                                                 const uint32_t *__restrict__ job_bits)
{
    __shared__ __attribute__((aligned(16))) float s_seg[SDF_BLOCK * SDF_ENTRY];
    __shared__ float s_best[MULTI ? 16u * 64u : 1u];     // per-pixel minima of the region between blocks of segments
    __shared__ float s_xy[SDF_RW + SDF_RH];     // the region's 64 sample abscissae, then its 16 sample ordinates
    __shared__ uint32_t s_bits[2u * SDF_RH];    // the sign bits of the region's 16 rows, two words each (bit x % 32 of word 2 y + x / 32: pixel (X0 + x, Y0 + y))
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[SDF_RW * SDF_RH];   // the region's finished bytes, stored at the end as whole 64-byte runs
    // Workgroups go to the 8 XCDs round-robin by their index, and each XCD has its own L2.  Logical index = the workgroups
    // of XCD 0 first, then XCD 1's, ...: regions next to each other in x — which read the same lines of the sign bit plane
    // and the same segments — run on ONE XCD, one after the other, and find those lines in its L2.
    uint32_t bid = blockIdx.x;
    {
        const uint32_t n = gridDim.x, per = n / 8u, rem = n % 8u, xcd = bid % 8u, slot = bid / 8u;
        bid = xcd * per + min(xcd, rem) + slot;     // (XCD x holds per + [x < rem] workgroups)
    }
    const uint32_t rxi = bid % regions_x; bid /= regions_x;
    const uint32_t ryi = bid % regions_y;
    const uint32_t jidx = bid / regions_y;
    const Job job = jobs[jidx];
    const uint32_t X0 = rxi * SDF_RW, Y0 = ryi * SDF_RH;
    if (X0 >= job.w || Y0 >= job.h) return;
    const uint32_t lane = threadIdx.x;
    const float off = phase_center ? 0.5f : 0.0f;
    const float scale = job.scale;
    // sample coordinates exactly as renderGlyph's (render_glyph.zig:26-27): lane l holds the abscissa of the region's
    // column l, lanes 0 .. 15 also the ordinate of its row l
    const float xs = ((float)(job.min_x + (int32_t)(X0 + lane)) + off) / scale;
    const float ys = ((float)(job.max_y - (int32_t)(Y0 + (lane & 15u))) - off) / scale;
    s_xy[lane] = xs;
    if (lane < SDF_RH) s_xy[SDF_RW + lane] = ys;
    // the sign: one word of the job's bit plane per region row (the plane's rows are ceil(w / 32) words; X0 is a
    // multiple of 32) — or, for a job of the general kernel, the byte that kernel left in the output
    const uint32_t jb = job_bits ? job_bits[jidx] : 0xffffffffu;
    const bool use_bits = jb != 0xffffffffu;                    // (wave-uniform)
    if (use_bits) {
        // lane = 16 pixels of one region row (row lane >> 2, quarter lane & 3): the tile starts as the saturated encoding —
        // 255 inside / 0 outside, what every pixel farther than the reach from the outline keeps
        // (one plane per 256-pixel column of the cell, h rows of 8 words each: fr_win1.hip; X0 is a multiple of 64.  The four
        // regions that share these 16 x 32 bytes run on the same XCD, back to back: see the block index below)
        const uint32_t row = lane >> 2, half = (lane >> 1) & 1u;
        const uint32_t wbits = (Y0 + row < job.h) ? bits[(size_t)jb + ((size_t)(X0 / 256u) * job.h + (Y0 + row)) * 8u + (X0 % 256u) / 32u + half] : 0u;
        if ((lane & 1u) == 0u) s_bits[2u * row + half] = wbits;
        const uint32_t b16 = (wbits >> (16u * (lane & 1u))) & 0xffffu;
        // four sign bits -> four bytes 0 / 255: 1 + 2^7 + 2^14 + 2^21 puts bit i at bit 8 i
        auto spread = [](uint32_t nib) -> uint32_t { return (((nib & 15u) * 0x00204081u) & 0x01010101u) * 255u; };
        *reinterpret_cast<uint4 *>(s_tile + 16u * lane) = make_uint4(spread(b16), spread(b16 >> 4), spread(b16 >> 8), spread(b16 >> 12));
    }
    // A set of sample points is tested as the disc about its box's centre: radius = half the diagonal, plus the
    // pixel test's slack at its farthest corner (see there), plus a margin for the roundings of this very sum.
    // Both maps are monotone in the pixel index, so the corners are the first and last column / row.
    auto disc = [](float xa, float xb, float ya, float yb, float &cx, float &cy, float &rad) {
        cx = 0.5f * (xa + xb); cy = 0.5f * (ya + yb);
        const float hx = 0.5f * (xb - xa), hy = 0.5f * (yb - ya);
        rad = __builtin_sqrtf(hx * hx + hy * hy) * 1.001f + 0.5f + 2.0e-6f * (fabsf(cx) + fabsf(cy) + hx + hy);
    };
    auto bcast = [](float v, uint32_t src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)src)); };
    float rcx, rcy, rrad;
    disc(bcast(xs, 0), bcast(xs, SDF_RW - 1u), bcast(ys, SDF_RH - 1u), bcast(ys, 0), rcx, rcy, rrad);
    // 8 pixels is where the encoding saturates; 2 % and one font unit of slack cover every rounding.
    // cull == 0 (ctx option "sdf_cull", tests): no segment is ever dropped or skipped
    const float reach = cull ? 8.0f / scale * 1.02f + 1.0f : 3.0e+37f;
    const float reach2 = reach * reach;
    const uint32_t s0 = job_seg[2u * (size_t)jidx], s1 = s0 + job_seg[2u * (size_t)jidx + 1u];      // (loaded beside the job, not after it)
    uint32_t touched = 0u;                       // quads that hold minima in s_best (wave-uniform, one bit per quad)
    bool have_quads = false;
    float qcx = 0.0f, qcy = 0.0f, qhx = 0.0f, qhy = 0.0f, qpad = 0.0f;   // the sample box of quad (lane & 15): centre, half sides, slack
    for (uint32_t base = s0; base < s1; base += SDF_BLOCK) {
        const bool last = base + SDF_BLOCK >= s1;
        // ---- stage the block's segments that are within reach of the region
        const uint32_t s = base + lane;
        bool keep = false;
        float c[6];
        Obb ob = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (lane < SDF_BLOCK && s < s1) {
            const uint32_t *w = reinterpret_cast<const uint32_t *>(seg_pts + 6u * (size_t)s);
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
            c[0] = (float)(int16_t)(w0 & 0xffffu); c[1] = (float)(int16_t)(w0 >> 16);
            c[2] = (float)(int16_t)(w1 & 0xffffu); c[3] = (float)(int16_t)(w1 >> 16);
            c[4] = (float)(int16_t)(w2 & 0xffffu); c[5] = (float)(int16_t)(w2 >> 16);
            ob = make_obb(c);
            keep = !cull || obb_gap2(c[0], c[1], ob.tx, ob.ty, ob.alo, ob.ahi, ob.blo, ob.bhi, rcx, rcy, rrad) <= reach2;
        }
        const unsigned long long km = __ballot(keep);
        const uint32_t n = (uint32_t)__popcll(km);
        if (n == 0u && !(last && touched)) continue;
        __syncthreads();                         // (one wave: the previous block's readers are done with the list; s_xy is written)
        if (keep) {
            float *e = s_seg + SDF_ENTRY * __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
            const float ax = c[2] - c[0], ay = c[3] - c[1];                                     // A
            const float bx = c[0] - 2.0f * c[2] + c[4], by = c[1] - 2.0f * c[3] + c[5];         // second difference
            *reinterpret_cast<float4 *>(e) = make_float4(c[0], c[1], ax, ay);
            *reinterpret_cast<float4 *>(e + 4) = make_float4(ob.tx, ob.ty, ob.alo, ob.ahi);
            *reinterpret_cast<float4 *>(e + 8) = make_float4(ob.blo, ob.bhi, bx, by);
            float bxk[8], byk[8];
#pragma unroll
            for (int k = 1; k <= 8; ++k) {
                const float t = (float)k * 0.125f;
                bxk[k - 1] = c[0] + 2.0f * t * ax + t * t * bx;
                byk[k - 1] = c[1] + 2.0f * t * ay + t * t * by;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                *reinterpret_cast<float4 *>(e + 12 + 4 * k) = make_float4(bxk[4 * k], bxk[4 * k + 1], bxk[4 * k + 2], bxk[4 * k + 3]);
                *reinterpret_cast<float4 *>(e + 20 + 4 * k) = make_float4(byk[4 * k], byk[4 * k + 1], byk[4 * k + 2], byk[4 * k + 3]);
            }
        }
        if (!have_quads) {                       // (the first block that keeps anything)
            have_quads = true;
            const uint32_t qc = (lane & (SDF_QX - 1u)) * 8u, qr = SDF_RW + ((lane & 15u) / SDF_QX) * 8u;
            const float xa = s_xy[qc], xb = s_xy[qc + 7u], ya = s_xy[qr + 7u], yb = s_xy[qr];
            qcx = 0.5f * (xa + xb); qcy = 0.5f * (ya + yb);
            qhx = 0.5f * (xb - xa) * 1.001f; qhy = 0.5f * (yb - ya) * 1.001f;
            qpad = 0.5f + 2.0e-6f * (fabsf(qcx) + fabsf(qcy) + qhx + qhy);                  // >= the pixel test's slack anywhere in the quad
        }
        __syncthreads();
        // ---- which list entries can reach which quad: lane (g, q) = (lane >> 4, lane & 15) tests the entries
        // g J .. g J + J - 1 against quad q; bit j of `cm` <-> entry g J + j
        const uint32_t J = (n + 3u) >> 2;
        uint32_t cm = 0u;
        for (uint32_t j = 0; j < J; ++j) {
            const uint32_t k = (lane >> 4) * J + j;
            if (k < n) {
                const float *e = s_seg + SDF_ENTRY * k;
                const float2 e0 = *reinterpret_cast<const float2 *>(e);
                const float4 e1 = *reinterpret_cast<const float4 *>(e + 4);
                const float2 e2 = *reinterpret_cast<const float2 *>(e + 8);
                // the quad's box seen in the segment's frame: its half extents along tau and n (a box in that frame
                // that holds the quad's sample points; tighter than the disc about them unless the segment is diagonal)
                const float atx = fabsf(e1.x), aty = fabsf(e1.y);
                const float pa = qhx * atx + qhy * aty + qpad, pb = qhx * aty + qhy * atx + qpad;
                const float wx = qcx - e0.x, wy = qcy - e0.y;
                const float fa = wx * e1.x + wy * e1.y, fb = wy * e1.x - wx * e1.y;
                const float da = fmaxf(fmaxf(e1.z - fa, fa - e1.w) - pa, 0.0f), db = fmaxf(fmaxf(e2.x - fb, fb - e2.y) - pb, 0.0f);
                const bool cand = !cull || da * da + db * db <= reach2;
                cm |= (cand ? 1u : 0u) << j;
            }
        }
        // ---- the region's quads
        for (uint32_t q = 0; q < 16u; ++q) {
            const uint32_t QX = X0 + (q % SDF_QX) * 8u, QY = Y0 + (q / SDF_QX) * 8u;
            if (QX >= job.w || QY >= job.h) continue;
            unsigned long long mask = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)cm, (int)q)
                                    | (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)cm, (int)(q + 16u)) << J
                                    | (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)cm, (int)(q + 32u)) << (2u * J)
                                    | (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)cm, (int)(q + 48u)) << (3u * J);
            const bool was = (touched >> q) & 1u;
            if (!mask && !(last && was)) continue;
            const uint32_t x = QX + (lane & 7u), y = QY + (lane >> 3);
            const bool valid = x < job.w && y < job.h;
            const float qx = s_xy[(q % SDF_QX) * 8u + (lane & 7u)], qy = s_xy[SDF_RW + (q / SDF_QX) * 8u + (lane >> 3)];
            // A computed curve point B(t) - q can leave the box by a few roundings of its three-term sum (each <= half
            // an ulp of a magnitude <= 2^18 for i16 points: < 2^-6 font units, far less relative to a distant sample),
            // and the box's own frame is rounded too (make_obb).  The box the pixel looks at is therefore grown by
            // `slack`, so that "the computed distance is >= the box distance" holds for the ROUNDED distance too and a
            // skipped segment can never have produced a smaller value — whatever order the segments come in.
            const float slack = 0.5f + 2.0e-6f * (fabsf(qx) + fabsf(qy));
            float best = (MULTI && was) ? s_best[q * 64u + lane] : 3.402823466e+38f;
            while (mask) {
                const uint32_t k = (uint32_t)__builtin_ctzll(mask);
                mask &= mask - 1ull;
                const float *e = s_seg + SDF_ENTRY * k;
                const float2 e0 = *reinterpret_cast<const float2 *>(e);
                const float4 e1 = *reinterpret_cast<const float4 *>(e + 4);
                const float2 e2 = *reinterpret_cast<const float2 *>(e + 8);
                const float g2 = obb_gap2(e0.x, e0.y, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, qx, qy, slack);
                const bool active = valid && !(cull && (g2 >= best || g2 > reach2));
                if (!__any(active)) continue;
#ifdef FR_SDF_STATS
                {
                    const unsigned long long am = __ballot(active);
                    if (lane == 0) { atomicAdd(&g_sdf_stats[0], 1ull); atomicAdd(&g_sdf_stats[1], (unsigned long long)__popcll(am)); }
                }
#endif
#if defined(FR_SDF_ABLATE) && FR_SDF_ABLATE == 1
                const float d2 = cull == 7 ? seg_dist2(e, qx, qy) : g2 + 1.0f;                  // timing-only: no distance evaluation
#else
                const float d2 = seg_dist2(e, qx, qy, active);
#endif
                if (active && d2 < best) best = d2;
            }
            if (MULTI && !last) {
                s_best[q * 64u + lane] = best;
                touched |= 1u << q;
                continue;
            }
            if (!valid) continue;
            const uint32_t trow = (q / SDF_QX) * 8u + (lane >> 3), tcol = (q % SDF_QX) * 8u + (lane & 7u);     // my pixel in the region
            uint8_t *px = use_bits ? s_tile + trow * SDF_RW + tcol
                                   : out + ((size_t)job.out_y + y) * out_stride + job.out_x + x;
            // the sign pass: winding != 0 at this sample
            const bool inside = use_bits ? ((s_bits[2u * trow + (tcol >> 5)] >> (tcol & 31u)) & 1u) != 0u : *px != 0;
            float d = __builtin_sqrtf(best) * scale;
            if (!inside) d = -d;
            float v = 16.0f * d + 128.0f;
            v = floorf(v + 0.5f);
            v = fminf(fmaxf(v, 0.0f), 255.0f);
            *px = (uint8_t)v;
        }
    }
    if (!use_bits) return;                       // (the sign bytes already ARE the saturated encoding of every other pixel)
    // ---- the region leaves as 16 runs of 64 bytes (lane = 16 bytes: row lane >> 2, quarter lane & 3), clipped to the cell
    __syncthreads();                             // (one wave; the tile is complete)
    {
        const uint32_t row = lane >> 2, x0 = X0 + 16u * (lane & 3u), y = Y0 + row;
        if (y < job.h && x0 < job.w) {
            const uint4 v = *reinterpret_cast<const uint4 *>(s_tile + 16u * lane);
            uint8_t *dst = out + ((size_t)job.out_y + y) * out_stride + job.out_x + x0;
            if (x0 + 16u <= job.w) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                typedef u32x4 u32x4_u __attribute__((aligned(4)));      // (global memory takes the unaligned 16-byte store: as c4_store16)
                const u32x4 w = {v.x, v.y, v.z, v.w};
                *reinterpret_cast<u32x4_u *>(dst) = w;
            } else {
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (uint32_t i = 0; i < 16u; ++i)
                    if (x0 + i < job.w) dst[i] = (uint8_t)(w4[i >> 2] >> (8u * (i & 3u)));
            }
        }
    }
}

hipError_t launch_sdf(const RenderArgs &a, uint32_t max_w, uint32_t max_h, uint32_t max_seg, int cull, hipStream_t stream)
{
    if (a.n_jobs == 0 || max_w == 0 || max_h == 0) return hipSuccess;
    const uint32_t rx = (max_w + SDF_RW - 1u) / SDF_RW, ry = (max_h + SDF_RH - 1u) / SDF_RH;
    const size_t grid = (size_t)a.n_jobs * rx * ry;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    if (max_seg > SDF_BLOCK)
        hipLaunchKernelGGL(sdf_kernel<true>, dim3((uint32_t)grid), dim3(64), 0, stream, a.jobs, a.job_seg, a.seg_pts,
                           reinterpret_cast<uint8_t *>(a.out), a.out_stride, rx, ry, a.phase_center, cull, a.bits, a.job_bits);
    else
        hipLaunchKernelGGL(sdf_kernel<false>, dim3((uint32_t)grid), dim3(64), 0, stream, a.jobs, a.job_seg, a.seg_pts,
                           reinterpret_cast<uint8_t *>(a.out), a.out_stride, rx, ry, a.phase_center, cull, a.bits, a.job_bits);
    return hipGetLastError();
}

}  // namespace fr
