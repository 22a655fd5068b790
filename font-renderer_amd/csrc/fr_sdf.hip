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
//                render_glyph.zig:29,35-73), - outside; taken from the byte render_kernel's 1-sample
//                coverage left in the output (255 / 0) right before this kernel
//   encoding   : u8 = clamp(floor(128 + 16*d + 0.5), 0, 255)   (8 pixels of range either side)
//
// Shape: one workgroup per 16x16-pixel tile of a cell, one lane per pixel; the glyph's control
// points are staged through LDS as f32 (256 segments at a time) and read back as broadcasts.
// The encoding saturates 8 pixels from the outline, so a segment whose control-point box (the curve
// lies inside it) is farther than that from the whole tile cannot change any of the tile's bytes: it
// is dropped while staging (LDS counter compaction; the minimum does not depend on the order).  The
// distance the kernel computes is the distance to SOME point of the curve, hence >= the box distance:
// every dropped segment would have produced a value past the clamp.
#include "fr_device.hpp"

namespace fr {

__device__ __forceinline__ float seg_dist2(float p0x, float p0y, float p1x, float p1y, float p2x, float p2y,
                                           float qx, float qy)
{
    const float ax = p1x - p0x, ay = p1y - p0y;                         // A
    const float bx = p0x - 2.0f * p1x + p2x, by = p0y - 2.0f * p1y + p2y;   // second difference
    float best = 3.402823466e+38f, bt = 0.0f;
    for (int k = 0; k <= 8; ++k) {
        const float t = (float)k * 0.125f;
        const float x = p0x + 2.0f * t * ax + t * t * bx - qx;
        const float y = p0y + 2.0f * t * ay + t * t * by - qy;
        const float d2 = x * x + y * y;
        if (d2 < best) { best = d2; bt = t; }
    }
    const float lo = fmaxf(bt - 0.125f, 0.0f), hi = fminf(bt + 0.125f, 1.0f);
    float t = bt;
    for (int it = 0; it < 4; ++it) {
        const float x = p0x + 2.0f * t * ax + t * t * bx - qx;
        const float y = p0y + 2.0f * t * ay + t * t * by - qy;
        const float dx = 2.0f * (ax + t * bx), dy = 2.0f * (ay + t * by);
        const float f = x * dx + y * dy;
        const float fp = dx * dx + dy * dy + 2.0f * (x * bx + y * by);
        if (fp > 0.0f) t = t - f / fp;
        t = fminf(fmaxf(t, lo), hi);
    }
    {
        const float x = p0x + 2.0f * t * ax + t * t * bx - qx;
        const float y = p0y + 2.0f * t * ay + t * t * by - qy;
        const float d2 = x * x + y * y;
        if (d2 < best) best = d2;
    }
    return best;
}

__global__ __launch_bounds__(256) void sdf_kernel(const Job *__restrict__ jobs, const int16_t *__restrict__ pts,
                                                  const uint32_t *__restrict__ seg_p0,
                                                  const uint32_t *__restrict__ glyph_seg_start,
                                                  const uint32_t *__restrict__ glyph_rec_count,
                                                  const Rec *__restrict__ recs, uint8_t *__restrict__ out,
                                                  uint64_t out_stride, uint32_t tiles_x, uint32_t tiles_y,
                                                  int phase_center, int cull)
{
    __shared__ float s_seg[256][10];            // p0, p1, p2 and the control-point box (x0, x1, y0, y1)
    __shared__ uint32_t s_n;
    uint32_t bid = blockIdx.x;
    const uint32_t tx = bid % tiles_x; bid /= tiles_x;
    const uint32_t ty = bid % tiles_y;
    const uint32_t jidx = bid / tiles_y;
    const Job job = jobs[jidx];
    if (tx * 16u >= job.w || ty * 16u >= job.h) return;
    // each wave takes an 8x8 quarter of the tile (not a 16x4 strip): the per-pixel cull below only saves
    // work when all 64 lanes agree, and a compact square agrees more often
    const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
    const uint32_t x = tx * 16u + (wv & 1u) * 8u + (ln & 7u), y = ty * 16u + (wv >> 1) * 8u + (ln >> 3);
    const bool valid = x < job.w && y < job.h;
    const float off = phase_center ? 0.5f : 0.0f;
    const float qx = ((float)(job.min_x + (int32_t)x) + off) / job.scale;       // render_glyph.zig:26
    const float qy = ((float)(job.max_y - (int32_t)y) - off) / job.scale;       // :27
    const uint32_t s0 = glyph_seg_start[job.glyph], s1 = glyph_seg_start[job.glyph + 1];
    // the tile's sample points span [tqx0, tqx1] x [tqy0, tqy1] (font units; the maps are monotone)
    const float tqx0 = ((float)(job.min_x + (int32_t)(tx * 16u)) + off) / job.scale;
    const float tqx1 = ((float)(job.min_x + (int32_t)(tx * 16u + 15u)) + off) / job.scale;
    const float tqy1 = ((float)(job.max_y - (int32_t)(ty * 16u)) - off) / job.scale;
    const float tqy0 = ((float)(job.max_y - (int32_t)(ty * 16u + 15u)) - off) / job.scale;
    // 8 pixels is where the encoding saturates; 2 % and one font unit of slack cover every rounding.
    // cull == 0 (ctx option "sdf_cull", tests): no segment is ever dropped or skipped
    const float reach = cull ? 8.0f / job.scale * 1.02f + 1.0f : 3.0e+37f;
    // A computed curve point B(t) - q can leave the control-point box by a few roundings of its three-term sum
    // (each <= half an ulp of a magnitude <= 2^18 for i16 points: < 2^-6 font units, far less relative to a
    // distant sample).  The boxes the culls look at are therefore grown by `slack`, so that "the computed
    // distance is >= the box distance" holds for the ROUNDED distance too and a skipped segment can never have
    // produced a smaller value — whatever order the segments were staged in.
    const float slack = 0.25f + 1.0e-6f * (fabsf(qx) + fabsf(qy));
    float best = 3.402823466e+38f;
    for (uint32_t base = s0; base < s1; base += 256u) {
        if (threadIdx.x == 0) s_n = 0u;
        __syncthreads();
        const uint32_t s = base + threadIdx.x;
        if (s < s1) {
            const int16_t *p = pts + 2u * (size_t)seg_p0[s];
            float c[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) c[k] = (float)p[k];
            const float hx0 = fminf(fminf(c[0], c[2]), c[4]), hx1 = fmaxf(fmaxf(c[0], c[2]), c[4]);
            const float hy0 = fminf(fminf(c[1], c[3]), c[5]), hy1 = fmaxf(fmaxf(c[1], c[3]), c[5]);
            const float gx = fmaxf(fmaxf(hx0 - tqx1, tqx0 - hx1), 0.0f), gy = fmaxf(fmaxf(hy0 - tqy1, tqy0 - hy1), 0.0f);
            if (!cull || gx * gx + gy * gy <= reach * reach) {
                const uint32_t slot = atomicAdd(&s_n, 1u);
#pragma unroll
                for (int k = 0; k < 6; ++k) s_seg[slot][k] = c[k];
                s_seg[slot][6] = hx0; s_seg[slot][7] = hx1; s_seg[slot][8] = hy0; s_seg[slot][9] = hy1;
            }
        }
        __syncthreads();
        const uint32_t n = s_n;
        if (valid)
            for (uint32_t k = 0; k < n; ++k) {
                // per pixel the same argument: the computed distance is >= the distance to the segment's box,
                // so a box no nearer than the best so far (or than the saturation reach) changes nothing
                const float gx = fmaxf(fmaxf(s_seg[k][6] - qx, qx - s_seg[k][7]) - slack, 0.0f);
                const float gy = fmaxf(fmaxf(s_seg[k][8] - qy, qy - s_seg[k][9]) - slack, 0.0f);
                const float g2 = gx * gx + gy * gy;
                if (cull && (g2 >= best || g2 > reach * reach)) continue;
                const float d2 = seg_dist2(s_seg[k][0], s_seg[k][1], s_seg[k][2], s_seg[k][3], s_seg[k][4], s_seg[k][5], qx, qy);
                if (d2 < best) best = d2;
            }
        __syncthreads();
    }
    if (!valid) return;
    uint8_t *px = out + ((size_t)job.out_y + y) * out_stride + job.out_x + x;
    const bool inside = *px != 0;              // render_kernel<COVERAGE_U8, 1>: winding != 0 at this sample
    float d = (s1 > s0) ? __builtin_sqrtf(best) * job.scale : 3.402823466e+38f;
    if (!inside) d = -d;
    float v = 16.0f * d + 128.0f;
    v = floorf(v + 0.5f);
    v = fminf(fmaxf(v, 0.0f), 255.0f);
    *px = (uint8_t)v;
}

hipError_t launch_sdf(const RenderArgs &a, const int16_t *pts, const uint32_t *seg_p0, uint32_t max_w,
                      uint32_t max_h, int cull, hipStream_t stream)
{
    if (a.n_jobs == 0 || max_w == 0 || max_h == 0) return hipSuccess;
    const uint32_t tiles_x = (max_w + 15u) / 16u, tiles_y = (max_h + 15u) / 16u;
    const size_t grid = (size_t)a.n_jobs * tiles_x * tiles_y;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sdf_kernel, dim3((uint32_t)grid), dim3(256), 0, stream, a.jobs, pts, seg_p0,
                       a.glyph_seg_start, a.glyph_rec_count, a.recs, reinterpret_cast<uint8_t *>(a.out),
                       a.out_stride, tiles_x, tiles_y, a.phase_center, cull);
    return hipGetLastError();
}

}  // namespace fr
