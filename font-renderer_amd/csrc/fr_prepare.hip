// fr_prepare.hip — stand-alone per-glyph-set precompute of the root records (fr_records.hpp).
// Used at glyph-set creation (the records serve the render kernel's over-full rows and the SDF sign) and, per render,
// for glyphs too large for the render kernel's in-LDS record build.
#include "fr_records.hpp"

namespace fr {

// One wave64 per glyph.  Lane l builds candidate record c = base + l, where
// candidate 2s / 2s+1 are the t+ / t- roots of segment s (2s alone for a == 0);
// survivors (records whose bracket is not provably empty) are compacted with a ballot + prefix popcount into the glyph's slice
// [2*seg_start[g], ...) of the record arrays.
__global__ __launch_bounds__(64) void prepare_kernel(const int16_t *__restrict__ pts,
                                                     const uint32_t *__restrict__ seg_p0,
                                                     const uint32_t *__restrict__ glyph_seg_start,
                                                     const uint32_t *__restrict__ glyph_list,
                                                     uint32_t n_glyphs,
                                                     Rec *__restrict__ out_recs,
                                                     uint32_t *__restrict__ glyph_rec_count)
{
    if (blockIdx.x >= n_glyphs) return;
    // the whole set (list == nullptr), or only the listed glyphs (a plan's glyphs of more than 128 segments)
    const uint32_t g = glyph_list ? glyph_list[blockIdx.x] : blockIdx.x;
    const uint32_t s0 = glyph_seg_start[g], s1 = glyph_seg_start[g + 1];
    const uint32_t n_cand = 2u * (s1 - s0);
    const uint32_t lane = threadIdx.x;
    const size_t out_base = 2u * (size_t)s0;
    uint32_t n_out = 0;
    for (uint32_t base = 0; base < n_cand; base += 64u) {
        const uint32_t c = base + lane;
        bool valid = false;
        Rec r;
        if (c < n_cand) valid = build_record(pts + 2u * (size_t)seg_p0[s0 + (c >> 1)], c & 1u, r);
        const unsigned long long m = __ballot(valid);
        if (valid) {
            const uint32_t pos = n_out + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            out_recs[out_base + pos] = r;
        }
        n_out += (uint32_t)__popcll(m);
    }
    if (lane == 0) glyph_rec_count[g] = n_out;
}

void launch_prepare(const int16_t *pts, const uint32_t *seg_p0, const uint32_t *glyph_seg_start,
                    const uint32_t *glyph_list, uint32_t n_glyphs, Rec *out_recs, uint32_t *glyph_rec_count,
                    hipStream_t stream)
{
    if (n_glyphs == 0) return;
    hipLaunchKernelGGL(prepare_kernel, dim3(n_glyphs), dim3(64), 0, stream, pts, seg_p0,
                       glyph_seg_start, glyph_list, n_glyphs, out_recs, glyph_rec_count);
}

}  // namespace fr
