// fr_api.hip — host side of libfr_raster.so: the C ABI of include/fr_raster.h.
//
// Owns device memory (glyph tables, root records, job tables), validates what the
// caller hands over, and launches the kernels of fr_prepare.hip / fr_render.hip /
// fr_exact.hip on the context's HIP stream.  No CPU rasterization path exists here:
// without a usable gfx950 device every compute entry point returns FR_E_HIP.
#include "../../include/fr_raster.h"
#include "fr_device.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <algorithm>
#include <dlfcn.h>
#include <vector>

namespace fr {
void launch_prepare(const int16_t *, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t, Rec *,
                    uint32_t *, hipStream_t);
hipError_t launch_render(const RenderArgs &, int mode, int n, hipStream_t, bool launch = true, char *name = nullptr, size_t name_cap = 0);
uint32_t render_wg_waves();
hipError_t launch_cov4(const RenderArgs &, uint32_t rec_cap, int ns, hipStream_t, bool launch = true, char *name = nullptr, size_t name_cap = 0);
hipError_t launch_win1(const RenderArgs &, int mode1, uint32_t rec_cap, hipStream_t, bool launch = true, char *name = nullptr, size_t name_cap = 0);
uint32_t cov4_wg_waves();
uint32_t cov4_max_segments();
hipError_t launch_sdf(const RenderArgs &, uint32_t, uint32_t, uint32_t max_seg, int cull, hipStream_t);
void launch_glyph_info(const int16_t *, const uint32_t *, const uint32_t *, uint32_t, int, uint8_t *,
                       uint8_t *, hipStream_t);
void launch_exact_winding(const int16_t *, const uint32_t *, const uint8_t *, const uint8_t *,
                          uint32_t, const int16_t *, uint64_t, uint32_t, int, int, int, int16_t *,
                          hipStream_t);
void launch_exact_cover(const int16_t *, uint32_t, uint32_t, uint32_t, uint8_t *, hipStream_t);
void launch_glyph_debug_color(const int16_t *, uint64_t, uint32_t, uint32_t, uint8_t *, hipStream_t);
}  // namespace fr

static_assert(sizeof(fr_job) == sizeof(fr::Job), "fr_job layout");
static_assert(sizeof(fr_job) == 32, "fr_job is 32 bytes");

// --------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
namespace fr {
int set_error(int code, const char *fmt, ...)          // for the other translation units (fr_font.cpp)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace fr
#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? FR_E_NOMEM : FR_E_HIP, "%s: %s", #expr,  \
                        hipGetErrorString(e_));                                              \
    } while (0)

struct fr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    uint32_t kmax = 32;          // crossings kept per sample row (register array: 8, 16 or 32) before the direct-sum fallback
    uint32_t strip_px = 256;     // column strip width, pixels (multiple of 16, <= 256)
    uint32_t fuse_prepare = 1;   // build root records inside the render kernel when every glyph has <= 128 segments
    uint32_t lds_pad = 0;        // experiment knob: extra dynamic LDS bytes per workgroup (occupancy studies)
    uint32_t min_wgs = 2048;     // split a cell's bands over workgroups below this many workgroups
    uint32_t cov4 = 1;           // jobs take the fast kernels (cov4_kernel / win1_kernel) where they fit (fast_class); 0: all general
    uint32_t zero_copy = 0;      // fr_render_glyph: render small glyphs from / into pinned host memory directly (measured: no faster than two small copies; off)
    uint32_t sdf_cull = 1;       // FR_SDF_U8: drop segments that cannot change a tile / a pixel (exact; 0 = look at all, for tests)
    uint32_t overlap = 1;        // a plan's smaller launches run beside its largest one on a second stream: 0 never, 1 plans of >= 32 Mpixel, 2 always
    uint32_t graph = 0;          // 1: a plan's launches are captured into a hipGraph at its first render to a destination and replayed afterwards
    uint32_t opt_epoch = 0;      // bumped by every fr_ctx_set_option: a captured graph is only replayed under the options it was captured with
    hipStream_t aux = nullptr;   // that second stream and the fork / join events, created on first use
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // scratch of the single-glyph entry point (fr_render_glyph): one device arena and one host staging
    // buffer, grown on demand and reused across calls
    unsigned char *arena = nullptr;
    size_t arena_cap = 0;
    unsigned char *stage = nullptr;     // pinned host memory: [upload block | image coming back]
    size_t stage_cap = 0;
};

struct fr_glyphset {
    fr_ctx *ctx = nullptr;
    uint32_t n_glyphs = 0, n_contours = 0, n_seg = 0, max_seg_per_glyph = 0;
    uint64_t n_points = 0;
    int16_t *d_pts = nullptr, *d_seg_pts = nullptr;
    uint32_t *d_seg_p0 = nullptr, *d_seg_prev = nullptr, *d_glyph_seg_start = nullptr, *d_rec_count = nullptr;
    fr::Rec *d_recs = nullptr;
    std::vector<uint32_t> h_glyph_seg_start;    // host copy: plans attach each job's segment range to it
    std::vector<uint32_t> h_root_bound;         // per glyph: candidate roots the vertex rule cannot discard (>= live records)
    std::vector<uint32_t> h_ray_bound;          // per glyph: estimated maximum of the crossings of one horizontal ray
};

struct fr_plan {
    fr_ctx *ctx = nullptr;
    const fr_glyphset *gs = nullptr;
    fr::Job *d_jobs = nullptr;
    uint32_t *d_job_seg = nullptr;     // [n_jobs][2]: first segment and segment count of the job's glyph
    uint32_t *d_bits = nullptr;        // FR_SDF_U8: the sign bit planes of the fast kernels' jobs (one bit per pixel: fr_win1.hip)
    uint32_t *d_job_bits = nullptr;    // and each job's first word in them (0xffffffff: a general-kernel job)
    uint32_t *d_large = nullptr;       // distinct glyphs of more than 128 segments among the jobs: their records are
    uint32_t n_large = 0;              // rebuilt by prepare_kernel before every render (the others: inside the render kernel)
    uint32_t n_jobs = 0;
    // jobs cov4_kernel / win1_kernel take (fr_cov4.hip, fr_win1.hip): the first n_fast entries of d_jobs / d_job_seg, grouped
    // into `parts` — one launch each, by strip width (64 / 128 / 256 pixels, from the job's own width) and by the record
    // slots the glyph needs (128 / 256 / 512); the general kernel renders the other n_jobs - n_fast
    uint32_t n_fast = 0;
    struct Part { uint32_t first, cnt, wlog, rec_cap, bands, strips; uint64_t pixels; };
    std::vector<Part> parts;
    int fast_ns = 0;                   // samples per axis of the fast kernels' jobs (4 / 2: cov4_kernel, 1: win1_kernel)
    uint32_t gen_bands = 0, gen_strips = 0;
    bool gen_uniform = false;
    fr_raster_params params{};
    uint32_t bands = 0, strips = 0, strip_w = 0, max_w = 0, max_h = 0;
    bool uniform = false;        // see fr_plan_create
    uint64_t pixels = 0, need_cols = 0, need_rows = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // option "graph": the launches of one render as an instantiated hipGraph, and what it was captured for
    hipGraphExec_t gexec = nullptr;
    void *g_out = nullptr;
    size_t g_stride = 0, g_rows = 0;
    uint32_t g_epoch = 0;
};

template <class T> static void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }

extern "C" {

int fr_abi_version(void) { return FR_ABI_VERSION; }
const char *fr_last_error(void) { return g_err; }
#ifndef FR_BUILD_ID
#error "FR_BUILD_ID comes from the Makefile (a hash of the sources)"
#endif
const char *fr_build_id(void) { return FR_BUILD_ID; }

int fr_ctx_create(int device, void *hip_stream, fr_ctx **out)
{
    if (!out) return fail(FR_E_INVALID, "fr_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(FR_E_HIP, "fr_ctx_create: no HIP device (%s); this library has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "count = 0");
    if (device < 0 || device >= n) return fail(FR_E_INVALID, "fr_ctx_create: device %d of %d", device, n);
    HIP_TRY(hipSetDevice(device));
    fr_ctx *c = new (std::nothrow) fr_ctx;
    if (!c) return fail(FR_E_NOMEM, "fr_ctx_create: host allocation");
    c->device = device;
    if (hip_stream) {
        c->stream = static_cast<hipStream_t>(hip_stream);
    } else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(FR_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        c->owns_stream = true;
    }
    *out = c;
    return FR_OK;
}

void fr_ctx_destroy(fr_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int fr_ctx_sync(fr_ctx *ctx)
{
    if (!ctx) return fail(FR_E_INVALID, "fr_ctx_sync: ctx is NULL");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FR_OK;
}

int fr_ctx_set_option(fr_ctx *ctx, const char *key, int64_t value)
{
    if (!ctx || !key) return fail(FR_E_INVALID, "fr_ctx_set_option: NULL argument");
    ++ctx->opt_epoch;
    if (!strcmp(key, "graph")) { ctx->graph = value ? 1u : 0u; return FR_OK; }
    if (!strcmp(key, "kmax")) {
        if (value < 1 || value > 128) return fail(FR_E_INVALID, "kmax must be in [1,128]");
        ctx->kmax = (uint32_t)value;
        return FR_OK;
    }
    if (!strcmp(key, "strip_px")) {
        if (value < 16 || value > 256 || (value % 16)) return fail(FR_E_INVALID, "strip_px must be a multiple of 16 in [16,256]");
        ctx->strip_px = (uint32_t)value;
        return FR_OK;
    }
    if (!strcmp(key, "fuse_prepare")) { ctx->fuse_prepare = value ? 1u : 0u; return FR_OK; }
    if (!strcmp(key, "lds_pad")) { ctx->lds_pad = (uint32_t)value; return FR_OK; }
    if (!strcmp(key, "cov4")) { ctx->cov4 = value ? 1u : 0u; return FR_OK; }
    if (!strcmp(key, "sdf_cull")) { ctx->sdf_cull = value ? 1u : 0u; return FR_OK; }
    if (!strcmp(key, "overlap")) {
        if (value < 0 || value > 2) return fail(FR_E_INVALID, "overlap must be 0 (never), 1 (plans of >= 32 Mpixel) or 2 (always)");
        ctx->overlap = (uint32_t)value;
        return FR_OK;
    }
    if (!strcmp(key, "zero_copy")) { ctx->zero_copy = value ? 1u : 0u; return FR_OK; }
    if (!strcmp(key, "min_wgs")) {
        if (value < 1 || value > (1 << 24)) return fail(FR_E_INVALID, "min_wgs out of range");
        ctx->min_wgs = (uint32_t)value;
        return FR_OK;
    }
    return fail(FR_E_INVALID, "fr_ctx_set_option: unknown key '%s'", key);
}

// ---- glyph tables --------------------------------------------------------
// Builds, per curve, the point index of its p0 and of the previous curve's p0 in the
// same contour (wrapping to the last curve: render_glyph.zig:126-127).
static int flatten_segments(const uint32_t *contour_start, uint32_t n_contours, uint64_t *n_points,
                            std::vector<uint32_t> &seg_p0, std::vector<uint32_t> &seg_prev,
                            std::vector<uint32_t> *contour_seg_start)
{
    if (n_contours && !contour_start) return fail(FR_E_INVALID, "contour_start is NULL");
    uint64_t np = n_contours ? contour_start[n_contours] : 0;
    if (n_contours && contour_start[0] != 0) return fail(FR_E_INVALID, "contour_start[0] must be 0");
    if (contour_seg_start) contour_seg_start->assign(1, 0u);
    for (uint32_t c = 0; c < n_contours; ++c) {
        if (contour_start[c + 1] < contour_start[c]) return fail(FR_E_INVALID, "contour_start not monotone at %u", c);
        const uint32_t len = contour_start[c + 1] - contour_start[c];
        // points.len = 2*curves + 1 (Glyph.zig:23); an even length would index past the
        // slice in the reference (render_glyph.zig:42)
        if (len != 0 && (len & 1u) == 0) return fail(FR_E_INVALID, "contour %u has even length %u", c, len);
        const uint32_t curves = len / 2;                                    // render_glyph.zig:38
        for (uint32_t k = 0; k < curves; ++k) {
            seg_p0.push_back(contour_start[c] + 2 * k);
            seg_prev.push_back(contour_start[c] + (k != 0 ? 2 * k - 2 : len - 3));
        }
        if (contour_seg_start) contour_seg_start->push_back((uint32_t)seg_p0.size());
    }
    *n_points = np;
    return FR_OK;
}

}  // extern "C"

// Upper bound of the root records a render can keep for a glyph (segments [s0, s1)): the two candidates of a segment
// minus those build_record_rows (fr_records.hpp) discards without looking at a cell — a == 0: one root, none if
// p2y == p0y (render_glyph.zig:49-50); else the far-side root when t_v = B/a >= 1 and the near-side root when t_v < 0.
static uint32_t glyph_root_bound(const int16_t *points_xy, const uint32_t *seg_p0, uint32_t s0, uint32_t s1)
{
    uint32_t nb = 0;
    for (uint32_t sgi = s0; sgi < s1; ++sgi) {
        const int16_t *q = points_xy + 2u * (size_t)seg_p0[sgi];
        const int32_t p0y = q[1], p1y = q[3], p2y = q[5];
        const int32_t a = p0y - 2 * p1y + p2y, b = p0y - p1y;
        if (a == 0) { nb += (p2y != p0y) ? 1u : 0u; continue; }
        const int64_t ba = (int64_t)b * a;
        const bool tv_lt0 = ba < 0, tv_ge1 = a > 0 ? b >= a : b <= a;
        nb += (tv_ge1 ? 0u : 1u) + (tv_lt0 ? 0u : 1u);
    }
    return nb;
}

// Estimate of the most crossings one horizontal ray can have with a glyph: a sweep over the segments' y extents (the
// control points bound the curve), counted once between the heights of its ends and twice where it overshoots them.
// Glyphs that stay at or under 16 take the instance that keeps 16 crossings per sample row in registers (plan_classify).
// `ev` is scratch: (2 y + [closing], +-weight) — openings sort before closings at one y.
static uint32_t glyph_ray_bound(const int16_t *points_xy, const uint32_t *seg_p0, uint32_t s0, uint32_t s1,
                                std::vector<std::pair<int32_t, int32_t>> &ev)
{
    ev.clear();
    for (uint32_t sgi = s0; sgi < s1; ++sgi) {
        const int16_t *q = points_xy + 2u * (size_t)seg_p0[sgi];
        const int32_t p0y = q[1], p1y = q[3], p2y = q[5];
        // between the heights of its two ends a quadratic is met once; where it overshoots them (towards the
        // control point: the vertex lies inside) twice, and not at all between the ends' heights on that side
        const int32_t clo = std::min(p0y, p2y), chi = std::max(p0y, p2y);
        // (half-open at the ends' heights, as the reference's own t in [0, 1) is: two segments that meet at a
        // vertex are not both counted there.  An estimate that steers jobs, not a proof: a row that does hold
        // more than the instance keeps takes the exact direct sum)
        ev.emplace_back(2 * clo, 1);
        ev.emplace_back(2 * chi, -1);
        // (the vertex overshoots the nearer end by at most half of what the control point does)
        if (p1y > chi) { ev.emplace_back(2 * chi, 2); ev.emplace_back(2 * (chi + (p1y - chi + 1) / 2) + 1, -2); }
        if (p1y < clo) { ev.emplace_back(2 * (clo - (clo - p1y + 1) / 2), 2); ev.emplace_back(2 * clo, -2); }
    }
    std::sort(ev.begin(), ev.end());
    int32_t cur = 0, best = 0;
    for (const auto &e : ev) { cur += e.second; best = std::max(best, cur); }
    return (uint32_t)best;
}

// Which kernel renders a job.  cov4_kernel (ns x ns samples, ns in {2, 4}) and win1_kernel (one sample per pixel)
// take cells of ANY width and height — renderGlyph's own image size (render_glyph.zig:14-19) included — up to 2048
// sample rows, of glyphs with <= 384 segments and <= 512 root records the vertex rule cannot discard: in strips of
// 64 / 128 / 256 pixels chosen from the job's own width (a 47 x 45 image does not pay for 256 columns) and bands of
// 64 sample rows, the last strip and band clipped at the cell's border.  Everything else takes the general
// render_kernel.  -> 0 (general) or 1 + 4 (wlog - 2) + record class (0: <= 128 slots and <= 16 crossings per ray
// estimated, 1: <= 256 slots, 2: <= 512 (<= 384 segments), 3: <= 1024 (<= 768 segments: two workgroups per CU)).
enum { FAST_RC = 4, FAST_CLASSES = 3 * FAST_RC };     // record classes (128 / 256 / 512 / 1024 slots) x strip widths (64 / 128 / 256)
struct FastRule {
    int ns = 0;             // samples per axis on the fast kernels (0: this plan has no fast kernel)
    uint32_t wlog_max = 0;  // widest strip the context allows (option "strip_px")
};
static FastRule fast_rule(const fr_ctx *ctx, const fr_raster_params *params)
{
    FastRule r;
    const int n = params->samples_per_axis;
    const bool one = params->mode == FR_WINDING_I16 || params->mode == FR_GRAY_DEBUG || params->mode == FR_MASK_NONZERO ||
                     (params->mode == FR_COVERAGE_U8 && n == 1) || params->mode == FR_SDF_U8;   // (SDF: its sign pass)
    r.wlog_max = ctx->strip_px >= 256u ? 4u : (ctx->strip_px >= 128u ? 3u : (ctx->strip_px >= 64u ? 2u : 0u));
    if (ctx->cov4 && r.wlog_max) r.ns = one ? 1 : ((params->mode == FR_COVERAGE_U8 && (n == 4 || n == 2)) ? n : 0);
    return r;
}
static int fast_class(const FastRule &R, uint32_t w, uint32_t h, uint32_t nsg, uint32_t root_bound, uint32_t ray_bound)
{
    if (!R.ns || w == 0 || h == 0 || (uint64_t)h * (uint32_t)R.ns > 2048u) return 0;     // (12-bit sample-row fields)
    if (nsg > fr::cov4_max_segments() || root_bound > 1024u) return 0;
    const uint32_t wl = std::min(w <= 64u ? 2u : (w <= 128u ? 3u : 4u), R.wlog_max);
    const int rc = (nsg <= 256u && root_bound <= 128u && ray_bound <= 16u) ? 0 : ((nsg <= 256u && root_bound <= 256u) ? 1 :
                   ((nsg <= 384u && root_bound <= 512u) ? 2 : 3));
    return 1 + FAST_RC * (int)(wl - 2u) + rc;
}
// classes of fewer than FAST_PART_MIN jobs move up into the next class that has jobs: same strip width and more record
// slots first, then wider strips with at least as many record slots (class c = 3 (wlog - 2) + record class; cls[j] = c + 1)
enum { FAST_PART_MIN = 64 };
static void merge_small_classes(uint32_t counts[FAST_CLASSES], uint8_t *cls, uint32_t n_jobs)
{
    int remap[FAST_CLASSES];
    bool any = false;
    for (int c = 0; c < FAST_CLASSES; ++c) {
        remap[c] = c;
        if (counts[c] == 0 || counts[c] >= (uint32_t)FAST_PART_MIN) continue;
        const int w = c / FAST_RC, r = c % FAST_RC;
        int target = -1;
        for (int w2 = w; w2 < 3 && target < 0; ++w2)
            for (int r2 = (w2 == w ? r + 1 : r); r2 < FAST_RC; ++r2)
                if (counts[FAST_RC * w2 + r2]) { target = FAST_RC * w2 + r2; break; }
        if (target < 0) continue;
        counts[target] += counts[c];          // (the target may be small itself: it is looked at later in this loop)
        counts[c] = 0;
        remap[c] = target;
        any = true;
    }
    if (!any) return;
    for (int c = 0; c < FAST_CLASSES; ++c) {             // chains: a -> b -> c
        int t = remap[c];
        while (remap[t] != t) t = remap[t];
        remap[c] = t;
    }
    for (uint32_t j = 0; j < n_jobs; ++j)
        if (cls[j]) cls[j] = (uint8_t)(remap[cls[j] - 1] + 1);
}

// the fast jobs of `order` (already grouped by class, `counts[c]` jobs of class c + 1) -> the plan's launches
static void make_parts(fr_plan *p, const fr_job *sorted_jobs, const uint32_t counts[FAST_CLASSES], int ns)
{
    p->parts.clear();
    p->fast_ns = ns;
    if (ns <= 0) return;                                              // (no fast kernel in this plan)
    const uint32_t prb = ns == 1 ? 16u : 64u / (uint32_t)ns;          // pixel rows of a band
    uint32_t first = 0;
    for (int c = 0; c < FAST_CLASSES; ++c) {
        if (!counts[c]) continue;
        fr_plan::Part pt{};
        pt.first = first; pt.cnt = counts[c]; pt.wlog = 2u + (uint32_t)(c / FAST_RC); pt.rec_cap = 128u << (c % FAST_RC);
        const uint32_t sw = 16u << pt.wlog;
        for (uint32_t q = first; q < first + counts[c]; ++q) {
            pt.bands = std::max(pt.bands, (sorted_jobs[q].h + prb - 1u) / prb);
            pt.strips = std::max(pt.strips, (sorted_jobs[q].w + sw - 1u) / sw);
            pt.pixels += (uint64_t)sorted_jobs[q].w * sorted_jobs[q].h;
        }
        p->parts.push_back(pt);
        first += counts[c];
    }
}

extern "C" {

void fr_glyphset_destroy(fr_glyphset *gs)
{
    if (!gs) return;
    (void)hipSetDevice(gs->ctx->device);
    (void)hipStreamSynchronize(gs->ctx->stream);
    dfree(gs->d_pts); dfree(gs->d_seg_pts); dfree(gs->d_seg_p0); dfree(gs->d_seg_prev); dfree(gs->d_glyph_seg_start);
    dfree(gs->d_rec_count); dfree(gs->d_recs);
    delete gs;
}

int fr_glyphset_prepare(fr_glyphset *gs)
{
    if (!gs) return fail(FR_E_INVALID, "fr_glyphset_prepare: NULL");
    HIP_TRY(hipSetDevice(gs->ctx->device));
    fr::launch_prepare(gs->d_pts, gs->d_seg_p0, gs->d_glyph_seg_start, nullptr, gs->n_glyphs, gs->d_recs,
                       gs->d_rec_count, gs->ctx->stream);
    HIP_TRY(hipGetLastError());
    return FR_OK;
}

int fr_glyphset_create(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                       uint32_t n_contours, const uint32_t *glyph_start, uint32_t n_glyphs,
                       fr_glyphset **out)
{
    if (!ctx || !out) return fail(FR_E_INVALID, "fr_glyphset_create: NULL argument");
    *out = nullptr;
    if (n_glyphs && !glyph_start) return fail(FR_E_INVALID, "glyph_start is NULL");
    if (n_glyphs && (glyph_start[0] != 0 || glyph_start[n_glyphs] != n_contours))
        return fail(FR_E_INVALID, "glyph_start must run from 0 to n_contours");
    std::vector<uint32_t> seg_p0, seg_prev, cseg;
    uint64_t np = 0;
    int rc = flatten_segments(contour_start, n_contours, &np, seg_p0, seg_prev, &cseg);
    if (rc) return rc;
    if (np && !points_xy) return fail(FR_E_INVALID, "points_xy is NULL");
    if (seg_p0.size() > 0x7fffffffull) return fail(FR_E_UNSUPPORTED, "too many segments");
    std::vector<uint32_t> gseg(n_glyphs + 1, 0u);
    for (uint32_t g = 0; g < n_glyphs; ++g) {
        if (glyph_start[g + 1] < glyph_start[g] || glyph_start[g + 1] > n_contours)
            return fail(FR_E_INVALID, "glyph_start not monotone at %u", g);
        gseg[g + 1] = cseg[glyph_start[g + 1]];
    }
    uint32_t max_seg = 0;
    for (uint32_t g = 0; g < n_glyphs; ++g) max_seg = std::max(max_seg, gseg[g + 1] - gseg[g]);
    std::vector<uint32_t> root_bound(n_glyphs, 0u), ray_bound(n_glyphs, 0u);
    {
        std::vector<std::pair<int32_t, int32_t>> ev;
        for (uint32_t g = 0; g < n_glyphs; ++g) {
            root_bound[g] = glyph_root_bound(points_xy, seg_p0.data(), gseg[g], gseg[g + 1]);
            ray_bound[g] = glyph_ray_bound(points_xy, seg_p0.data(), gseg[g], gseg[g + 1], ev);
        }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    fr_glyphset *gs = new (std::nothrow) fr_glyphset;
    if (!gs) return fail(FR_E_NOMEM, "fr_glyphset_create: host allocation");
    gs->ctx = ctx; gs->n_glyphs = n_glyphs; gs->n_contours = n_contours;
    gs->n_seg = (uint32_t)seg_p0.size(); gs->n_points = np; gs->max_seg_per_glyph = max_seg;
    const size_t nseg1 = gs->n_seg ? gs->n_seg : 1, np1 = np ? np : 1;
#define GS_TRY(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            fr_glyphset_destroy(gs);                                                          \
            return fail(e_ == hipErrorOutOfMemory ? FR_E_NOMEM : FR_E_HIP, "%s: %s", #expr,   \
                        hipGetErrorString(e_));                                               \
        }                                                                                     \
    } while (0)
    GS_TRY(hipMalloc(&gs->d_pts, np1 * 2 * sizeof(int16_t) + 16));
    GS_TRY(hipMalloc(&gs->d_seg_pts, nseg1 * 12 + 16));
    GS_TRY(hipMalloc(&gs->d_seg_p0, nseg1 * 4));
    GS_TRY(hipMalloc(&gs->d_seg_prev, nseg1 * 4));
    GS_TRY(hipMalloc(&gs->d_glyph_seg_start, ((size_t)n_glyphs + 1) * 4));
    GS_TRY(hipMalloc(&gs->d_rec_count, ((size_t)n_glyphs + 1) * 4));
    GS_TRY(hipMalloc(&gs->d_recs, 2 * nseg1 * sizeof(fr::Rec)));
    hipStream_t st = ctx->stream;
    if (np) GS_TRY(hipMemcpyAsync(gs->d_pts, points_xy, np * 2 * sizeof(int16_t), hipMemcpyHostToDevice, st));
    std::vector<int16_t> seg_pts((size_t)gs->n_seg * 6);
    for (size_t sgi = 0; sgi < gs->n_seg; ++sgi) memcpy(&seg_pts[6 * sgi], points_xy + 2u * (size_t)seg_p0[sgi], 12);
    if (gs->n_seg) {
        GS_TRY(hipMemcpyAsync(gs->d_seg_pts, seg_pts.data(), (size_t)gs->n_seg * 12, hipMemcpyHostToDevice, st));
        GS_TRY(hipMemcpyAsync(gs->d_seg_p0, seg_p0.data(), (size_t)gs->n_seg * 4, hipMemcpyHostToDevice, st));
        GS_TRY(hipMemcpyAsync(gs->d_seg_prev, seg_prev.data(), (size_t)gs->n_seg * 4, hipMemcpyHostToDevice, st));
    }
    GS_TRY(hipMemcpyAsync(gs->d_glyph_seg_start, gseg.data(), ((size_t)n_glyphs + 1) * 4, hipMemcpyHostToDevice, st));
    gs->h_glyph_seg_start = gseg;
    gs->h_root_bound = root_bound;
    gs->h_ray_bound = ray_bound;
    GS_TRY(hipMemsetAsync(gs->d_rec_count, 0, ((size_t)n_glyphs + 1) * 4, st));
    fr::launch_prepare(gs->d_pts, gs->d_seg_p0, gs->d_glyph_seg_start, nullptr, n_glyphs, gs->d_recs,
                       gs->d_rec_count, st);
    GS_TRY(hipGetLastError());
    GS_TRY(hipStreamSynchronize(st));   // host vectors above die with this frame
#undef GS_TRY
    *out = gs;
    return FR_OK;
}

int fr_glyphset_stats(const fr_glyphset *gs, uint64_t *n_segments, uint64_t *n_records)
{
    if (!gs) return fail(FR_E_INVALID, "fr_glyphset_stats: NULL");
    HIP_TRY(hipSetDevice(gs->ctx->device));
    if (n_segments) *n_segments = gs->n_seg;
    if (n_records) {
        std::vector<uint32_t> cnt(gs->n_glyphs + 1);
        HIP_TRY(hipStreamSynchronize(gs->ctx->stream));
        HIP_TRY(hipMemcpy(cnt.data(), gs->d_rec_count, ((size_t)gs->n_glyphs) * 4, hipMemcpyDeviceToHost));
        uint64_t t = 0;
        for (uint32_t g = 0; g < gs->n_glyphs; ++g) t += cnt[g];
        *n_records = t;
    }
    return FR_OK;
}

// ---- plans ----------------------------------------------------------------
static int check_params(const fr_raster_params *p)
{
    if (!p) return fail(FR_E_INVALID, "params is NULL");
    if (p->mode < FR_WINDING_I16 || p->mode > FR_SDF_U8) return fail(FR_E_INVALID, "unknown mode %d", p->mode);
    const int n = p->samples_per_axis;
    if (p->mode == FR_COVERAGE_U8) {
        if (n != 1 && n != 2 && n != 4) return fail(FR_E_UNSUPPORTED, "samples_per_axis %d not in {1,2,4}", n);
    } else if (n != 1) {
        return fail(FR_E_INVALID, "samples_per_axis must be 1 for mode %d", p->mode);
    }
    if (p->sample_phase != FR_SAMPLE_CORNER && p->sample_phase != FR_SAMPLE_CENTER)
        return fail(FR_E_INVALID, "unknown sample_phase %d", p->sample_phase);
    return FR_OK;
}

void fr_plan_destroy(fr_plan *plan)
{
    if (!plan) return;
    (void)hipSetDevice(plan->ctx->device);
    (void)hipStreamSynchronize(plan->ctx->stream);
    dfree(plan->d_jobs); dfree(plan->d_job_seg); dfree(plan->d_large); dfree(plan->d_bits); dfree(plan->d_job_bits);
    if (plan->ev0) (void)hipEventDestroy(plan->ev0);
    if (plan->ev1) (void)hipEventDestroy(plan->ev1);
    if (plan->gexec) (void)hipGraphExecDestroy(plan->gexec);
    delete plan;
}

int fr_plan_create(fr_ctx *ctx, const fr_glyphset *gs, const fr_job *jobs, uint32_t n_jobs,
                   const fr_raster_params *params, fr_plan **out)
{
    if (!ctx || !gs || !out) return fail(FR_E_INVALID, "fr_plan_create: NULL argument");
    *out = nullptr;
    if (gs->ctx != ctx) return fail(FR_E_INVALID, "glyph set belongs to another context");
    int rc = check_params(params);
    if (rc) return rc;
    if (n_jobs && !jobs) return fail(FR_E_INVALID, "jobs is NULL");
    const uint32_t n = (uint32_t)params->samples_per_axis;
    uint32_t max_w = 0, max_h = 0;
    uint64_t pixels = 0, need_cols = 0, need_rows = 0;
    for (uint32_t j = 0; j < n_jobs; ++j) {
        const fr_job &jb = jobs[j];
        if (jb.glyph >= gs->n_glyphs) return fail(FR_E_INVALID, "job %u: glyph %u of %u", j, jb.glyph, gs->n_glyphs);
        if (!(jb.scale > 0.0f) || !std::isfinite(jb.scale)) return fail(FR_E_INVALID, "job %u: scale must be finite and > 0", j);
        // keeps every ray height, quotient and FMA residual of div_by_int inside the normal range
        // (the reference's own scale = u16 / u16 lies in [2^-16, 2^16])
        if (jb.scale < 9.5367431640625e-07f || jb.scale > 1048576.0f)
            return fail(FR_E_UNSUPPORTED, "job %u: scale outside [2^-20, 2^20]", j);
        if (jb.w > 65535u || jb.h > 65535u) return fail(FR_E_UNSUPPORTED, "job %u: cell larger than 65535", j);
        // sample coordinates must be exactly representable in binary32
        if (jb.min_x < -(1 << 22) || (int64_t)jb.min_x + jb.w > (1 << 22) || jb.max_y > (1 << 22) ||
            (int64_t)jb.max_y - jb.h < -(1 << 22))
            return fail(FR_E_UNSUPPORTED, "job %u: pixel coordinates beyond +-2^22", j);
        max_w = jb.w > max_w ? jb.w : max_w;
        max_h = jb.h > max_h ? jb.h : max_h;
        pixels += (uint64_t)jb.w * jb.h;
        need_cols = std::max<uint64_t>(need_cols, (uint64_t)jb.out_x + jb.w);
        need_rows = std::max<uint64_t>(need_rows, (uint64_t)jb.out_y + jb.h);
    }
    fr_plan *p = new (std::nothrow) fr_plan;
    if (!p) return fail(FR_E_NOMEM, "fr_plan_create: host allocation");
    p->ctx = ctx; p->gs = gs; p->n_jobs = n_jobs; p->params = *params;
    p->pixels = pixels; p->need_cols = need_cols; p->need_rows = need_rows;
    p->max_w = max_w; p->max_h = max_h;
    const uint32_t band = 64u / n;                                      // pixel rows per wave band
    const uint32_t cap_w = ctx->strip_px;                               // strip width cap, pixels
    // Per JOB: the fast kernels or the general one (fast_class above).  The job table is stored fast jobs first, grouped
    // by class — one launch per class that occurs.
    std::vector<uint32_t> order(n_jobs);
    uint32_t n_fast = 0;
    uint32_t counts[FAST_CLASSES] = {};
    const FastRule rule = fast_rule(ctx, params);
    {
        std::vector<uint8_t> cls(n_jobs);
        for (uint32_t j = 0; j < n_jobs; ++j) {
            const fr_job &jb = jobs[j];
            const uint32_t nsg = gs->h_glyph_seg_start[jb.glyph + 1] - gs->h_glyph_seg_start[jb.glyph];
            cls[j] = (uint8_t)fast_class(rule, jb.w, jb.h, nsg, gs->h_root_bound[jb.glyph], gs->h_ray_bound[jb.glyph]);
            if (cls[j]) { ++counts[cls[j] - 1]; ++n_fast; }
        }
        // A class with only a handful of jobs is not worth a launch of its own (a real font at renderGlyph's sizes: three or
        // four glyphs per odd class, each launch a few microseconds on the second stream): its jobs join the next class up
        // that exists — wider strips and / or more record slots render the same bytes (the stores are clipped, spare
        // record slots stay empty), only a little less efficiently.
        merge_small_classes(counts, cls.data(), n_jobs);
        uint32_t at[FAST_CLASSES + 1], run = 0;
        for (int c = 0; c < FAST_CLASSES; ++c) { at[c + 1] = run; run += counts[c]; }
        at[0] = run;                                                      // the general kernel's jobs go last
        for (uint32_t j = 0; j < n_jobs; ++j) order[at[cls[j]]++] = j;
    }
    p->n_fast = n_fast;
    // the general list: uniform = every strip of every job is full (w a multiple of the strip width) and every wave
    // band is full (h a multiple of 64 / n pixel rows) — atlas cells; the render kernel has instances for it
    p->uniform = n_jobs > n_fast;
    uint32_t gmax_w = 0, gmax_h = 0;
    for (uint32_t q = n_fast; q < n_jobs; ++q) {
        gmax_w = std::max(gmax_w, jobs[order[q]].w); gmax_h = std::max(gmax_h, jobs[order[q]].h);
    }
    uint32_t sw = (gmax_w + 15u) & ~15u;                                // the general kernel's strip width
    if (sw > cap_w) sw = cap_w;
    if (sw == 0) sw = 16;
    p->strip_w = sw;
    for (uint32_t q = n_fast; q < n_jobs; ++q) {
        const fr_job &jb = jobs[order[q]];
        if (jb.w == 0 || jb.h == 0 || jb.w % sw || jb.h % band) p->uniform = false;
    }
    p->gen_bands = gmax_h ? (gmax_h + band - 1) / band : 1;
    p->gen_strips = gmax_w ? (gmax_w + sw - 1) / sw : 1;
    p->bands = p->gen_bands; p->strips = p->gen_strips;
    std::vector<fr_job> sorted_jobs(n_jobs);
    for (uint32_t q = 0; q < n_jobs; ++q) sorted_jobs[q] = jobs[order[q]];
    make_parts(p, sorted_jobs.data(), counts, rule.ns);
    {
        bool too_many = (uint64_t)(n_jobs - n_fast) * p->gen_bands * p->gen_strips > 0x7fffffffull;
        for (const auto &pt : p->parts) too_many = too_many || (uint64_t)pt.cnt * pt.bands * pt.strips > 0x7fffffffull;
        if (too_many) {
            delete p;
            return fail(FR_E_UNSUPPORTED, "batch needs more than 2^31 workgroups; split it");
        }
    }
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess && n_jobs) e = hipMalloc(&p->d_jobs, (size_t)n_jobs * sizeof(fr::Job));
    if (e == hipSuccess && n_jobs)
        e = hipMemcpyAsync(p->d_jobs, sorted_jobs.data(), (size_t)n_jobs * sizeof(fr::Job), hipMemcpyHostToDevice, ctx->stream);
    // each job's segment range, next to the job: the render kernel starts on the glyph's points without a
    // dependent look-up through the glyph table
    std::vector<uint32_t> jseg((size_t)n_jobs * 2);
    for (uint32_t q = 0; q < n_jobs; ++q) {
        const uint32_t gl = sorted_jobs[q].glyph;
        jseg[2 * (size_t)q] = gs->h_glyph_seg_start[gl];
        jseg[2 * (size_t)q + 1] = gs->h_glyph_seg_start[gl + 1] - gs->h_glyph_seg_start[gl];
    }
    // glyphs too large for the in-kernel record build (> 128 segments) among the general kernel's jobs, each once
    std::vector<uint32_t> large;
    for (uint32_t q = n_fast; q < n_jobs; ++q)
        if (jseg[2 * (size_t)q + 1] > 128u) large.push_back(sorted_jobs[q].glyph);
    std::sort(large.begin(), large.end());
    large.erase(std::unique(large.begin(), large.end()), large.end());
    p->n_large = (uint32_t)large.size();
    if (e == hipSuccess && p->n_large) e = hipMalloc(&p->d_large, large.size() * 4);
    if (e == hipSuccess && p->n_large)
        e = hipMemcpyAsync(p->d_large, large.data(), large.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n_jobs) e = hipMalloc(&p->d_job_seg, (size_t)n_jobs * 8);
    if (e == hipSuccess && n_jobs)
        e = hipMemcpyAsync(p->d_job_seg, jseg.data(), (size_t)n_jobs * 8, hipMemcpyHostToDevice, ctx->stream);
    // FR_SDF_U8: the sign of a fast job travels as one bit per pixel in a plane of its own (win1_kernel's sign-bit mode
    // writes it, sdf_kernel reads it and is then the only writer of the output)
    std::vector<uint32_t> jbits;
    if (params->mode == FR_SDF_U8 && n_fast) {
        jbits.assign(n_jobs, 0xffffffffu);
        uint64_t words = 0;
        for (uint32_t q = 0; q < n_fast; ++q) {
            jbits[q] = (uint32_t)words;
            words += (uint64_t)((sorted_jobs[q].w + 255u) / 256u) * sorted_jobs[q].h * 8u;      // (one plane per 256-pixel column: h rows of 8 words)
            if (words >= 0xffffffffull) { e = hipErrorInvalidValue; break; }
        }
        if (e == hipSuccess) e = hipMalloc(&p->d_bits, (size_t)(words ? words : 1) * 4);
        if (e == hipSuccess) e = hipMalloc(&p->d_job_bits, (size_t)n_jobs * 4);
        if (e == hipSuccess) e = hipMemcpyAsync(p->d_job_bits, jbits.data(), (size_t)n_jobs * 4, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipEventCreate(&p->ev0);
    if (e == hipSuccess) e = hipEventCreate(&p->ev1);
    if (e != hipSuccess) {
        fr_plan_destroy(p);
        return fail(e == hipErrorOutOfMemory ? FR_E_NOMEM : FR_E_HIP, "fr_plan_create: %s", hipGetErrorString(e));
    }
    *out = p;
    return FR_OK;
}

uint64_t fr_plan_pixels(const fr_plan *plan) { return plan ? plan->pixels : 0; }

int fr_plan_stats(const fr_plan *plan, uint32_t *n_jobs_cov4, uint32_t *n_jobs_general)
{
    if (!plan) return fail(FR_E_INVALID, "fr_plan_stats: NULL");
    if (n_jobs_cov4) *n_jobs_cov4 = plan->n_fast;
    if (n_jobs_general) *n_jobs_general = plan->n_jobs - plan->n_fast;
    return FR_OK;
}

// the kernel instances a render of the plan launches, as rocprofv3 names them, with their job counts
int fr_plan_describe(const fr_plan *plan, char *buf, size_t cap)
{
    if (!plan || !buf || cap == 0) return fail(FR_E_INVALID, "fr_plan_describe: NULL argument");
    buf[0] = 0;
    size_t at = 0;
    auto add = [&](const char *name, uint32_t cnt) {
        const int k = snprintf(buf + at, cap - at, "%s%s x%u", at ? "; " : "", name, cnt);
        if (k > 0) at = std::min(cap - 1, at + (size_t)k);
    };
    fr::RenderArgs a{};
    a.kmax = plan->ctx->kmax;
    char name[96];
    const int pm = plan->params.mode;
    for (const auto &pt : plan->parts) {
        a.strip_w = 16u << pt.wlog;
        name[0] = 0;
        if (plan->fast_ns > 1) (void)fr::launch_cov4(a, pt.rec_cap, plan->fast_ns, nullptr, false, name, sizeof name);
        else (void)fr::launch_win1(a, pm == FR_WINDING_I16 ? 0 : (pm == FR_GRAY_DEBUG ? 1 : (pm == FR_SDF_U8 ? 3 : 2)), pt.rec_cap, nullptr, false, name, sizeof name);
        add(name, pt.cnt);
    }
    if (plan->n_jobs > plan->n_fast) {
        a.strip_w = plan->strip_w; a.uniform = plan->uniform ? 1u : 0u;
        name[0] = 0;
        if (pm == FR_SDF_U8) (void)fr::launch_render(a, FR_COVERAGE_U8, 1, nullptr, false, name, sizeof name);
        else (void)fr::launch_render(a, pm, plan->params.samples_per_axis, nullptr, false, name, sizeof name);
        add(name, plan->n_jobs - plan->n_fast);
    }
    if (pm == FR_SDF_U8 && plan->n_jobs) add(plan->gs->max_seg_per_glyph > 64u ? "fr::sdf_kernel<true>" : "fr::sdf_kernel<false>", plan->n_jobs);
    return FR_OK;
}

static int ensure_aux(fr_ctx *ctx)
{
    if (!ctx->aux) {
        HIP_TRY(hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    return FR_OK;
}

static int plan_check(fr_plan *plan, void *out_dev, size_t out_stride, size_t out_rows)
{
    if (!plan) return fail(FR_E_INVALID, "plan is NULL");
    if (plan->n_jobs == 0) return FR_OK;
    if (!out_dev) return fail(FR_E_INVALID, "out is NULL");
    if (plan->need_cols > out_stride || plan->need_rows > out_rows)
        return fail(FR_E_INVALID, "jobs need %llu x %llu elements, output is %zu x %zu",
                    (unsigned long long)plan->need_cols, (unsigned long long)plan->need_rows, out_stride, out_rows);
    // (the fast kernels address the rows of a wave band by 32-bit offsets from the band's base: 32 rows of the pitch)
    if (out_stride > ((size_t)1 << 26))
        return fail(FR_E_INVALID, "row pitch of %zu elements: at most 2^26", out_stride);
    return FR_OK;
}

// the launches of one render, issued on the context's stream(s)
static int plan_launch_direct(fr_plan *plan, void *out_dev, size_t out_stride, size_t out_rows)
{
    if (const int rc = plan_check(plan, out_dev, out_stride, out_rows)) return rc;
    if (plan->n_jobs == 0) return FR_OK;
    HIP_TRY(hipSetDevice(plan->ctx->device));
    const uint32_t n_fast = plan->n_fast, n_gen = plan->n_jobs - plan->n_fast;
    const bool sdf = plan->params.mode == FR_SDF_U8;
    fr::RenderArgs a;
    a.glyph_seg_start = plan->gs->d_glyph_seg_start;
    a.glyph_rec_count = plan->gs->d_rec_count;
    a.recs = plan->gs->d_recs;
    a.pts = plan->gs->d_pts;
    a.seg_p0 = plan->gs->d_seg_p0;
    a.seg_pts = plan->gs->d_seg_pts;
    a.job_bits = nullptr; a.bits = nullptr;
    // fused: the render kernel builds the records of every glyph of <= 128 segments (<= 256 candidate roots)
    // in LDS itself — decided per job inside the kernel; larger glyphs are staged from HBM
    a.fused = plan->ctx->fuse_prepare ? 1u : 0u;
    a.out = out_dev;
    a.out_stride = out_stride;
    a.strip_w = plan->strip_w;
    a.kmax = plan->ctx->kmax;
    a.phase_center = plan->params.sample_phase == FR_SAMPLE_CENTER ? 1 : 0;
    a.lds_pad = plan->ctx->lds_pad;
    a.nwin_log = 0; a.lds_region = 0; a.lds_rec_bytes = 0; a.lds_wave_bytes = 0; a.lds_tail = 0;
    // one workgroup walks all bands of its cell (cx table, job and records staged once)
    // unless the batch is too small to fill the chip: then split the bands over workgroups
    // (bands are wave bands of 64/n pixel rows; a workgroup's waves take them round-robin)
    auto split_bands = [&](uint32_t nw, uint32_t njobs, uint32_t bands, uint32_t strips) {
        uint32_t bpw = (bands + nw - 1u) / nw * nw;
        while (bpw > nw && (uint64_t)njobs * strips * ((bands + bpw - 1) / bpw) < plan->ctx->min_wgs) bpw = ((bpw / 2) + nw - 1u) / nw * nw;
        a.bands_per_wg = bpw;
        a.band_groups = (bands + bpw - 1) / bpw;
    };
    // A mixed plan: the smaller launches (a real font's few glyphs of many segments: the 512-record instance, the
    // general kernel) are short kernels with long critical paths — forked onto a second stream so that they run beside
    // the large one instead of before / after it (the jobs' cells are disjoint); joined before anything else touches the
    // output.
    fr_ctx *const ctx = plan->ctx;
    hipStream_t gst = ctx->stream;
    // (the largest fast launch stays on the context's stream; every other launch of the plan goes beside it)
    size_t big = 0;
    for (size_t i = 1; i < plan->parts.size(); ++i)
        if (plan->parts[i].pixels > plan->parts[big].pixels) big = i;
    const size_t n_launches = (n_gen ? 1u : 0u) + plan->parts.size();
    // (a small plan — a font at renderGlyph's own sizes for one font size: a few megapixels — is quicker launch after launch
    // on one stream than through a fork and a join: measured 0.038 vs 0.056 ms at 5.8 Mpixel, 0.373 vs 0.356 at 221 Mpixel)
    const bool forked = ctx->overlap && n_fast && n_launches > 1 && (ctx->overlap == 2u || plan->pixels >= ((uint64_t)32 << 20));
    if (forked) {
        if (const int rc = ensure_aux(ctx)) return rc;
        HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
        gst = ctx->aux;
    }
    // (whatever fails between the fork and the join: the context's stream still waits for the second one)
    auto launch_parts = [&]() -> int {
    // a render always starts from the glyph POINTS: inside the kernels (fused) or by re-running the stand-alone
    // precompute first, for the glyphs that need it
    if (n_gen && !a.fused)
        fr::launch_prepare(plan->gs->d_pts, plan->gs->d_seg_p0, plan->gs->d_glyph_seg_start, nullptr, plan->gs->n_glyphs,
                           plan->gs->d_recs, plan->gs->d_rec_count, gst);
    else if (n_gen && plan->n_large)
        fr::launch_prepare(plan->gs->d_pts, plan->gs->d_seg_p0, plan->gs->d_glyph_seg_start, plan->d_large, plan->n_large,
                           plan->gs->d_recs, plan->gs->d_rec_count, gst);
    if (n_gen) {
        a.jobs = plan->d_jobs + n_fast;
        a.job_seg = plan->d_job_seg + 2u * (size_t)n_fast;
        a.n_jobs = n_gen; a.bands = plan->gen_bands; a.strips = plan->gen_strips; a.uniform = plan->uniform ? 1u : 0u;
        split_bands(fr::render_wg_waves(), n_gen, plan->gen_bands, plan->gen_strips);
        // SDF, sign first: the 1-sample coverage (255 where the reference's winding is non-zero, same sample points)
        // lands in the output; the distance kernel reads it and overwrites it
        if (sdf) HIP_TRY(fr::launch_render(a, FR_COVERAGE_U8, 1, gst));
        else HIP_TRY(fr::launch_render(a, plan->params.mode, plan->params.samples_per_axis, gst));
    }
    for (size_t i = 0; i < plan->parts.size(); ++i) {
        // cov4_kernel / win1_kernel, one launch per (strip width, record slots) class that occurs in the plan
        const fr_plan::Part &pt = plan->parts[i];
        a.jobs = plan->d_jobs + pt.first;
        a.job_seg = plan->d_job_seg + 2u * (size_t)pt.first;
        a.n_jobs = pt.cnt; a.bands = pt.bands; a.strips = pt.strips; a.uniform = 1u;
        a.strip_w = 16u << pt.wlog;
        split_bands(fr::cov4_wg_waves(), pt.cnt, pt.bands, pt.strips);
        const int pm = plan->params.mode;
        hipStream_t pst = (forked && i != big) ? ctx->aux : ctx->stream;
        if (plan->fast_ns > 1) HIP_TRY(fr::launch_cov4(a, pt.rec_cap, plan->fast_ns, pst));
        else if (sdf && plan->d_bits) {
            // the sign pass of FR_SDF_U8: one bit per pixel into the job's own bit plane
            void *const keep = a.out;
            a.out = plan->d_bits; a.job_bits = plan->d_job_bits + pt.first;
            const hipError_t le = fr::launch_win1(a, 3, pt.rec_cap, pst);
            a.out = keep; a.job_bits = nullptr;
            HIP_TRY(le);
        }
        else HIP_TRY(fr::launch_win1(a, pm == FR_WINDING_I16 ? 0 : (pm == FR_GRAY_DEBUG ? 1 : 2), pt.rec_cap, pst));
    }
    return FR_OK;
    };
    const int rc_parts = launch_parts();
    if (forked) {
        HIP_TRY(hipEventRecord(ctx->ev_join, ctx->aux));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    }
    if (rc_parts) return rc_parts;
    if (sdf) {
        a.strip_w = plan->strip_w;
        a.bits = plan->d_bits; a.job_bits = plan->d_job_bits;
        a.jobs = plan->d_jobs;
        a.job_seg = plan->d_job_seg;
        a.n_jobs = plan->n_jobs;
        HIP_TRY(fr::launch_sdf(a, plan->max_w, plan->max_h, plan->gs->max_seg_per_glyph, (int)plan->ctx->sdf_cull, plan->ctx->stream));
    }
    return FR_OK;
}

// One render of a plan.  Option "graph": the same launches — the fork onto the second stream and the join included — are
// captured once per destination into a hipGraph and replayed with ONE hipGraphLaunch afterwards (the capture is redone when
// the destination or a context option changes).
static int plan_launch(fr_plan *plan, void *out_dev, size_t out_stride, size_t out_rows)
{
    if (const int rc = plan_check(plan, out_dev, out_stride, out_rows)) return rc;
    if (plan->n_jobs == 0) return FR_OK;
    fr_ctx *const ctx = plan->ctx;
    if (!ctx->graph) return plan_launch_direct(plan, out_dev, out_stride, out_rows);
    HIP_TRY(hipSetDevice(ctx->device));
    if (plan->gexec && (plan->g_out != out_dev || plan->g_stride != out_stride || plan->g_rows != out_rows || plan->g_epoch != ctx->opt_epoch)) {
        (void)hipGraphExecDestroy(plan->gexec);
        plan->gexec = nullptr;
    }
    if (!plan->gexec) {
        if (const int rc = ensure_aux(ctx)) return rc;        // (nothing is created while the capture is open)
        HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
        const int rc = plan_launch_direct(plan, out_dev, out_stride, out_rows);
        hipGraph_t g = nullptr;
        const hipError_t ce = hipStreamEndCapture(ctx->stream, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        HIP_TRY(ce);
        const hipError_t ie = hipGraphInstantiate(&plan->gexec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ie != hipSuccess) plan->gexec = nullptr;
        HIP_TRY(ie);
        plan->g_out = out_dev; plan->g_stride = out_stride; plan->g_rows = out_rows; plan->g_epoch = ctx->opt_epoch;
    }
    HIP_TRY(hipGraphLaunch(plan->gexec, ctx->stream));
    return FR_OK;
}

// ---- optional assembly over RCCL: bound at run time to the RCCL the host already uses (it created the communicator)
namespace {
using nccl_allgather_fn = int (*)(const void *, void *, size_t, int /* ncclDataType_t */, void * /* ncclComm_t */, hipStream_t);
using nccl_rank_fn = int (*)(void *, int *);
void *rccl_symbol(const char *name)
{
    if (void *p = dlsym(RTLD_DEFAULT, name)) return p;
    for (const char *lib : {"librccl.so", "librccl.so.1"})
        if (void *h = dlopen(lib, RTLD_NOW | RTLD_NOLOAD))            // only a library that is loaded already
            if (void *p = dlsym(h, name)) return p;
    return nullptr;
}
}  // namespace

int fr_allgather_bands(fr_ctx *ctx, void *nccl_comm, void *atlas_dev, size_t band_bytes)
{
    if (!ctx || !nccl_comm || !atlas_dev) return fail(FR_E_INVALID, "fr_allgather_bands: NULL argument");
    if (band_bytes == 0) return FR_OK;
    static const auto all_gather = reinterpret_cast<nccl_allgather_fn>(rccl_symbol("ncclAllGather"));
    static const auto user_rank = reinterpret_cast<nccl_rank_fn>(rccl_symbol("ncclCommUserRank"));
    if (!all_gather || !user_rank)
        return fail(FR_E_UNSUPPORTED, "fr_allgather_bands: no RCCL in this process (the host creates the communicator with it)");
    HIP_TRY(hipSetDevice(ctx->device));
    int rank = -1;
    if (user_rank(nccl_comm, &rank) != 0 || rank < 0) return fail(FR_E_INVALID, "fr_allgather_bands: ncclCommUserRank failed");
    const unsigned char *mine = static_cast<const unsigned char *>(atlas_dev) + (size_t)rank * band_bytes;
    const int rc = all_gather(mine, atlas_dev, band_bytes, 0 /* ncclInt8 / ncclChar */, nccl_comm, ctx->stream);
    if (rc != 0) return fail(FR_E_HIP, "fr_allgather_bands: ncclAllGather returned %d", rc);
    return FR_OK;
}

// Gather-to-root form of the same assembly (every peer has its own xGMI link to the root, so the root
// ingests up to 7 links' worth while no rank receives bytes it does not need — an all-gather moves W times the atlas):
// one RCCL group of point-to-point transfers — the root posts a receive per peer into that peer's slot of its atlas,
// every other rank one send of its own band.  root < 0: the all-gather above.
int fr_gather_bands(fr_ctx *ctx, void *nccl_comm, void *atlas_dev, size_t band_bytes, int root)
{
    if (root < 0) return fr_allgather_bands(ctx, nccl_comm, atlas_dev, band_bytes);
    if (!ctx || !nccl_comm || !atlas_dev) return fail(FR_E_INVALID, "fr_gather_bands: NULL argument");
    if (band_bytes == 0) return FR_OK;
    using p2p_fn = int (*)(void *, size_t, int, int, void *, hipStream_t);          // ncclSend / ncclRecv (buffer, count, type, peer, comm, stream)
    using group_fn = int (*)();
    using count_fn = int (*)(void *, int *);
    static const auto nsend = reinterpret_cast<p2p_fn>(rccl_symbol("ncclSend"));
    static const auto nrecv = reinterpret_cast<p2p_fn>(rccl_symbol("ncclRecv"));
    static const auto gstart = reinterpret_cast<group_fn>(rccl_symbol("ncclGroupStart"));
    static const auto gend = reinterpret_cast<group_fn>(rccl_symbol("ncclGroupEnd"));
    static const auto user_rank = reinterpret_cast<nccl_rank_fn>(rccl_symbol("ncclCommUserRank"));
    static const auto comm_count = reinterpret_cast<count_fn>(rccl_symbol("ncclCommCount"));
    if (!nsend || !nrecv || !gstart || !gend || !user_rank || !comm_count)
        return fail(FR_E_UNSUPPORTED, "fr_gather_bands: no RCCL in this process (the host creates the communicator with it)");
    HIP_TRY(hipSetDevice(ctx->device));
    int rank = -1, world = 0;
    if (user_rank(nccl_comm, &rank) != 0 || comm_count(nccl_comm, &world) != 0 || rank < 0 || world <= 0)
        return fail(FR_E_INVALID, "fr_gather_bands: ncclCommUserRank / ncclCommCount failed");
    if (root >= world) return fail(FR_E_INVALID, "fr_gather_bands: root %d of %d ranks", root, world);
    unsigned char *base = static_cast<unsigned char *>(atlas_dev);
    int rc = gstart();
    if (rc == 0) {
        if (rank == root) {
            for (int r = 0; r < world && rc == 0; ++r)
                if (r != root) rc = nrecv(base + (size_t)r * band_bytes, band_bytes, 0 /* ncclInt8 */, r, nccl_comm, ctx->stream);
        } else {
            rc = nsend(base + (size_t)rank * band_bytes, band_bytes, 0, root, nccl_comm, ctx->stream);
        }
        const int rc2 = gend();
        if (rc == 0) rc = rc2;
    }
    if (rc != 0) return fail(FR_E_HIP, "fr_gather_bands: RCCL returned %d", rc);
    return FR_OK;
}

int fr_plan_render(fr_plan *plan, void *out_dev, size_t out_stride, size_t out_rows)
{
    return plan_launch(plan, out_dev, out_stride, out_rows);
}

int fr_plan_render_timed(fr_plan *plan, void *out_dev, size_t out_stride, size_t out_rows, float *ms)
{
    if (!plan || !ms) return fail(FR_E_INVALID, "fr_plan_render_timed: NULL argument");
    HIP_TRY(hipSetDevice(plan->ctx->device));
    HIP_TRY(hipEventRecord(plan->ev0, plan->ctx->stream));
    int rc = plan_launch(plan, out_dev, out_stride, out_rows);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(plan->ev1, plan->ctx->stream));
    HIP_TRY(hipEventSynchronize(plan->ev1));
    HIP_TRY(hipEventElapsedTime(ms, plan->ev0, plan->ev1));
    return FR_OK;
}

int fr_render_batch(fr_ctx *ctx, const fr_glyphset *gs, const fr_job *jobs, uint32_t n_jobs,
                    const fr_raster_params *params, void *out_host, size_t out_stride, size_t out_rows)
{
    fr_plan *plan = nullptr;
    int rc = fr_plan_create(ctx, gs, jobs, n_jobs, params, &plan);
    if (rc) return rc;
    if (n_jobs == 0) { fr_plan_destroy(plan); return FR_OK; }
    if (!out_host) { fr_plan_destroy(plan); return fail(FR_E_INVALID, "out_host is NULL"); }
    const size_t esz = params->mode == FR_WINDING_I16 ? 2 : 1;
    const size_t bytes = out_stride * out_rows * esz;
    void *d_out = nullptr;
    hipError_t e = hipMalloc(&d_out, bytes ? bytes : 16);
    // pixels outside every job keep the caller's bytes: stage the buffer in first
    if (e == hipSuccess) e = hipMemcpyAsync(d_out, out_host, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        rc = plan_launch(plan, d_out, out_stride, out_rows);
        if (rc == FR_OK) {
            e = hipMemcpyAsync(out_host, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
    }
    if (d_out) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(d_out); }
    fr_plan_destroy(plan);
    if (rc) return rc;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? FR_E_NOMEM : FR_E_HIP, "fr_render_batch: %s", hipGetErrorString(e));
    return FR_OK;
}

// ---- renderGlyph drop-in ----------------------------------------------------
// render_glyph.zig:13-19, host arithmetic in binary32 exactly as written there.
int fr_render_glyph_dims(const int16_t box[4], uint16_t units_per_em, uint16_t font_size,
                         int16_t min_corner[2], int16_t max_corner[2], uint16_t *width,
                         uint16_t *height, float *scale_out)
{
    if (!box || !min_corner || !max_corner || !width || !height) return fail(FR_E_INVALID, "fr_render_glyph_dims: NULL argument");
    if (units_per_em == 0 || font_size == 0) return fail(FR_E_INVALID, "units_per_em and font_size must be > 0");
    const float scale = (float)font_size / (float)units_per_em;                    // :13
    const float b[4] = {(float)box[0] * scale, (float)box[1] * scale, (float)box[2] * scale, (float)box[3] * scale};   // :15
    const float lo0 = std::floor(b[0]), lo1 = std::floor(b[1]), hi0 = std::ceil(b[2]), hi1 = std::ceil(b[3]);
    if (lo0 < -32768.f || lo1 < -32768.f || hi0 > 32767.f || hi1 > 32767.f)
        return fail(FR_E_UNSUPPORTED, "scaled box leaves i16 (the reference's @intFromFloat would trap)");
    min_corner[0] = (int16_t)lo0; min_corner[1] = (int16_t)lo1;                    // :16
    max_corner[0] = (int16_t)hi0; max_corner[1] = (int16_t)hi1;                    // :17
    const int w = (int)max_corner[0] - min_corner[0] + 1, h = (int)max_corner[1] - min_corner[1] + 1;   // :18-19
    if (w < 1 || h < 1 || w > 32767 || h > 32767) return fail(FR_E_UNSUPPORTED, "image size leaves i16");
    *width = (uint16_t)w; *height = (uint16_t)h;
    if (scale_out) *scale_out = scale;
    return FR_OK;
}

// One glyph, one image — the reference's call shape.  No allocation per call: the glyph tables, the job and the
// output live in the context's device arena (grown on demand), filled by ONE host-to-device copy of a packed
// staging buffer; the render kernel builds the root records in LDS (prepare_kernel runs only for a glyph of
// more than 128 segments); the image comes back with one device-to-host copy.  The job covers the whole image,
// so the caller's buffer is never uploaded.
int fr_render_glyph(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                    uint32_t n_contours, const int16_t box[4], uint16_t units_per_em,
                    uint16_t font_size, int32_t mode, void *out_host)
{
    if (!ctx) return fail(FR_E_INVALID, "ctx is NULL");
    int16_t mn[2], mx[2];
    uint16_t w, h;
    float scale;
    int rc = fr_render_glyph_dims(box, units_per_em, font_size, mn, mx, &w, &h, &scale);
    if (rc) return rc;
    fr_raster_params prm{};
    prm.mode = mode; prm.samples_per_axis = 1; prm.sample_phase = FR_SAMPLE_CORNER;
    rc = check_params(&prm);
    if (rc) return rc;
    if (!out_host) return fail(FR_E_INVALID, "out_host is NULL");
    const uint32_t zero_start[1] = {0};
    std::vector<uint32_t> seg_p0, seg_prev;
    uint64_t np = 0;
    rc = flatten_segments(n_contours ? contour_start : zero_start, n_contours, &np, seg_p0, seg_prev, nullptr);
    if (rc) return rc;
    if (np && !points_xy) return fail(FR_E_INVALID, "points_xy is NULL");
    const uint32_t ns = (uint32_t)seg_p0.size();
    if (scale < 9.5367431640625e-07f || scale > 1048576.0f) return fail(FR_E_UNSUPPORTED, "scale outside [2^-20, 2^20]");
    fr_job jb{};
    jb.glyph = 0; jb.min_x = mn[0]; jb.max_y = mx[1]; jb.w = w; jb.h = h; jb.out_x = 0; jb.out_y = 0; jb.scale = scale;
    // arena layout (every part 16-byte aligned); the first `up` bytes are uploaded
    auto al = [](size_t v) { return (v + 15u) & ~(size_t)15; };
    const size_t esz = mode == FR_WINDING_I16 ? 2 : 1;
    const size_t o_pts = 0, o_p0 = al(o_pts + np * 4 + 16), o_spts = al(o_p0 + (size_t)ns * 4 + 4);
    const size_t o_gseg = al(o_spts + (size_t)ns * 12 + 16), o_job = al(o_gseg + 8), o_jseg = al(o_job + sizeof(fr_job));
    const size_t o_large = al(o_jseg + 8), up = al(o_large + 4);
    const size_t o_cnt = up, o_recs = al(o_cnt + 8), o_out = al(o_recs + (size_t)(ns ? ns : 1) * 2 * sizeof(fr::Rec));
    const size_t total = al(o_out + (size_t)w * h * esz);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t img_bytes = (size_t)w * h * esz;
    // Option "zero_copy" (off): the kernel reads the tables from, and writes the image to, the pinned staging block
    // itself (host memory mapped into the device's address space).  Measured on STIX 'A' in steady state: 34.9 us per
    // call against 34.7 us with the two small copies — no gain, so the copies stay the default.
    const bool zero_copy = ns <= 128u && mode != FR_SDF_U8 && total <= ((size_t)1 << 20) && ctx->zero_copy;
    if (!zero_copy && total > ctx->arena_cap) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->arena) { (void)hipFree(ctx->arena); ctx->arena = nullptr; ctx->arena_cap = 0; }
        const size_t cap = std::max<size_t>(total + total / 2, 1u << 20);
        HIP_TRY(hipMalloc(&ctx->arena, cap));
        ctx->arena_cap = cap;
    }
    const size_t stage_need = zero_copy ? total : up + img_bytes;
    if (stage_need > ctx->stage_cap) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->stage) { (void)hipHostFree(ctx->stage); ctx->stage = nullptr; ctx->stage_cap = 0; }
        const size_t cap = std::max<size_t>(stage_need * 2, 1u << 16);
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->stage), cap, hipHostMallocDefault));
        ctx->stage_cap = cap;
    }
    unsigned char *st = ctx->stage;
    memset(st, 0, up);
    if (np) memcpy(st + o_pts, points_xy, np * 4);
    if (ns) memcpy(st + o_p0, seg_p0.data(), (size_t)ns * 4);
    for (uint32_t sgi = 0; sgi < ns; ++sgi) memcpy(st + o_spts + 12u * (size_t)sgi, points_xy + 2u * (size_t)seg_p0[sgi], 12);
    const uint32_t gseg[2] = {0, ns}, jseg[2] = {0, ns}, large0[1] = {0};
    memcpy(st + o_gseg, gseg, 8); memcpy(st + o_job, &jb, sizeof jb); memcpy(st + o_jseg, jseg, 8); memcpy(st + o_large, large0, 4);
    unsigned char *A0 = zero_copy ? st : ctx->arena;
    if (!zero_copy) HIP_TRY(hipMemcpyAsync(A0, st, up, hipMemcpyHostToDevice, ctx->stream));
    // views of the arena dressed as a glyph set and a plan (nothing here owns memory: never destroyed)
    fr_glyphset gs;
    gs.ctx = ctx; gs.n_glyphs = 1; gs.n_contours = n_contours; gs.n_seg = ns; gs.max_seg_per_glyph = ns; gs.n_points = np;
    gs.d_pts = reinterpret_cast<int16_t *>(A0 + o_pts); gs.d_seg_pts = reinterpret_cast<int16_t *>(A0 + o_spts);
    gs.d_seg_p0 = reinterpret_cast<uint32_t *>(A0 + o_p0); gs.d_glyph_seg_start = reinterpret_cast<uint32_t *>(A0 + o_gseg);
    gs.d_rec_count = reinterpret_cast<uint32_t *>(A0 + o_cnt); gs.d_recs = reinterpret_cast<fr::Rec *>(A0 + o_recs);
    fr_plan pl;
    pl.ctx = ctx; pl.gs = &gs; pl.n_jobs = 1; pl.n_fast = 0; pl.params = prm;
    {
        // the same per-job rule as fr_plan_create: the image takes win1_kernel (64- / 128- / 256-pixel strips by its
        // own width) unless the glyph is too large for it
        const FastRule rule = fast_rule(ctx, &prm);
        std::vector<std::pair<int32_t, int32_t>> ev;
        const int cls = fast_class(rule, w, h, ns, glyph_root_bound(points_xy, seg_p0.data(), 0, ns),
                                   glyph_ray_bound(points_xy, seg_p0.data(), 0, ns, ev));
        if (cls && mode != FR_SDF_U8) {            // (one SDF image: the sign comes as a byte from the general kernel, no bit plane)
            uint32_t counts[FAST_CLASSES] = {};
            counts[cls - 1] = 1;
            pl.n_fast = 1;
            make_parts(&pl, &jb, counts, rule.ns);
        }
    }
    pl.d_jobs = reinterpret_cast<fr::Job *>(A0 + o_job); pl.d_job_seg = reinterpret_cast<uint32_t *>(A0 + o_jseg);
    pl.d_large = reinterpret_cast<uint32_t *>(A0 + o_large); pl.n_large = (ns > 128u && !pl.n_fast) ? 1u : 0u;
    pl.max_w = w; pl.max_h = h; pl.pixels = (uint64_t)w * h; pl.need_cols = w; pl.need_rows = h;
    uint32_t sw = ((uint32_t)w + 15u) & ~15u;
    if (sw > ctx->strip_px) sw = ctx->strip_px;
    pl.strip_w = sw;
    pl.bands = pl.gen_bands = ((uint32_t)h + 63u) / 64u;
    pl.strips = pl.gen_strips = ((uint32_t)w + sw - 1u) / sw;
    pl.uniform = false;
    int lrc = FR_OK;
    if (mode == FR_SDF_U8 || pl.n_large) {
        // the stand-alone records (float brackets): the staged path of a large glyph and the SDF's stand-alone users
        hipError_t e = hipMemsetAsync(gs.d_rec_count, 0, 8, ctx->stream);
        if (e != hipSuccess) lrc = fail(FR_E_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    if (lrc == FR_OK) lrc = plan_launch_direct(&pl, A0 + o_out, w, h);
    hipError_t e = hipSuccess;
    if (lrc == FR_OK && !zero_copy) e = hipMemcpyAsync(st + up, A0 + o_out, img_bytes, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (lrc == FR_OK && e == hipSuccess && e2 == hipSuccess) memcpy(out_host, zero_copy ? st + o_out : st + up, img_bytes);
    gs.d_pts = gs.d_seg_pts = nullptr; gs.d_seg_p0 = gs.d_glyph_seg_start = gs.d_rec_count = nullptr; gs.d_recs = nullptr;
    pl.d_jobs = nullptr; pl.d_job_seg = nullptr; pl.d_large = nullptr;
    if (lrc) return lrc;
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) return fail(FR_E_HIP, "fr_render_glyph: %s", hipGetErrorString(e));
    return FR_OK;
}

// ---- exact-integer path ---------------------------------------------------
struct ExactDev {
    int16_t *pts = nullptr;
    uint32_t *seg_p0 = nullptr, *seg_prev = nullptr;
    uint8_t *ctype = nullptr, *inc = nullptr;
    uint32_t n_seg = 0;
    int K = 1;
    ~ExactDev() { dfree(pts); dfree(seg_p0); dfree(seg_prev); dfree(ctype); dfree(inc); }
};

static int exact_setup(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                       uint32_t n_contours, ExactDev &d, int K = 1)
{
    if (!ctx) return fail(FR_E_INVALID, "ctx is NULL");
    std::vector<uint32_t> seg_p0, seg_prev;
    uint64_t np = 0;
    int rc = flatten_segments(contour_start, n_contours, &np, seg_p0, seg_prev, nullptr);
    if (rc) return rc;
    if (np && !points_xy) return fail(FR_E_INVALID, "points_xy is NULL");
    d.n_seg = (uint32_t)seg_p0.size();
    const size_t ns1 = d.n_seg ? d.n_seg : 1, np1 = np ? np : 1;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc(&d.pts, np1 * 4 + 16));
    HIP_TRY(hipMalloc(&d.seg_p0, ns1 * 4));
    HIP_TRY(hipMalloc(&d.seg_prev, ns1 * 4));
    HIP_TRY(hipMalloc(&d.ctype, ns1));
    HIP_TRY(hipMalloc(&d.inc, ns1));
    if (np) HIP_TRY(hipMemcpyAsync(d.pts, points_xy, np * 4, hipMemcpyHostToDevice, ctx->stream));
    if (d.n_seg) {
        HIP_TRY(hipMemcpyAsync(d.seg_p0, seg_p0.data(), (size_t)d.n_seg * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(d.seg_prev, seg_prev.data(), (size_t)d.n_seg * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    d.K = K;
    fr::launch_glyph_info(d.pts, d.seg_p0, d.seg_prev, d.n_seg, K, d.ctype, d.inc, ctx->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return FR_OK;
}

int fr_glyph_info_init(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                       uint32_t n_contours, uint8_t *curve_type, uint8_t *include_p0)
{
    ExactDev d;
    int rc = exact_setup(ctx, points_xy, contour_start, n_contours, d);
    if (rc) return rc;
    if (d.n_seg) {
        if (!curve_type || !include_p0) return fail(FR_E_INVALID, "output is NULL");
        HIP_TRY(hipMemcpy(curve_type, d.ctype, d.n_seg, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(include_p0, d.inc, d.n_seg, hipMemcpyDeviceToHost));
    }
    return FR_OK;
}

// cover_n > 0: the lattice is (cover_w * cover_n) x (cover_h * cover_n) points and out_host receives
// cover_w x cover_h u8 coverage values instead of the windings
static int exact_run(fr_ctx *ctx, ExactDev &d, const int16_t *query_xy, uint64_t n_query,
                     uint32_t lat_w, int lat_x0, int lat_y0, void *out_host,
                     uint32_t cover_w = 0, uint32_t cover_h = 0, uint32_t cover_n = 0)
{
    if (n_query == 0) return FR_OK;
    if (!out_host) return fail(FR_E_INVALID, "output is NULL");
    if (n_query > 0xffffffffull * 256ull) return fail(FR_E_UNSUPPORTED, "too many query points");
    int16_t *d_q = nullptr, *d_out = nullptr;
    hipError_t e = hipMalloc(&d_out, n_query * 2);
    if (e == hipSuccess && query_xy) e = hipMalloc(&d_q, n_query * 4);
    if (e == hipSuccess && query_xy) e = hipMemcpyAsync(d_q, query_xy, n_query * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        fr::launch_exact_winding(d.pts, d.seg_p0, d.ctype, d.inc, d.n_seg, d_q, n_query, lat_w, lat_x0, lat_y0, d.K, d_out, ctx->stream);
        e = hipGetLastError();
    }
    uint8_t *d_cov = nullptr;
    if (cover_n) {
        const size_t npx = (size_t)cover_w * cover_h;
        if (e == hipSuccess) e = hipMalloc(&d_cov, npx ? npx : 1);
        if (e == hipSuccess) {
            fr::launch_exact_cover(d_out, cover_w, cover_h, cover_n, d_cov, ctx->stream);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out_host, d_cov, npx, hipMemcpyDeviceToHost, ctx->stream);
    } else if (e == hipSuccess) {
        e = hipMemcpyAsync(out_host, d_out, n_query * 2, hipMemcpyDeviceToHost, ctx->stream);
    }
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = e2;
    dfree(d_q); dfree(d_out); dfree(d_cov);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? FR_E_NOMEM : FR_E_HIP, "exact winding: %s", hipGetErrorString(e));
    return FR_OK;
}

int fr_winding_in_glyph(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                        uint32_t n_contours, const int16_t *query_xy, uint32_t n_query,
                        int16_t *out_winding)
{
    if (n_query && !query_xy) return fail(FR_E_INVALID, "query_xy is NULL");
    ExactDev d;
    int rc = exact_setup(ctx, points_xy, contour_start, n_contours, d);
    if (rc) return rc;
    return exact_run(ctx, d, query_xy, n_query, 1, 0, 0, out_winding);
}

int fr_winding_lattice(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                       uint32_t n_contours, const int16_t box[4], int16_t *out_host)
{
    if (!box) return fail(FR_E_INVALID, "box is NULL");
    const int W = (int)box[2] - box[0] + 3, H = (int)box[3] - box[1] + 3;       // Image.zig:183
    if (W < 1 || H < 1) return fail(FR_E_INVALID, "empty box");
    ExactDev d;
    int rc = exact_setup(ctx, points_xy, contour_start, n_contours, d);
    if (rc) return rc;
    return exact_run(ctx, d, nullptr, (uint64_t)W * H, (uint32_t)W, box[0] - 1, box[3] + 1, out_host);
}

// 128-bit predicates hold dy*abxy^2 < 2^106 * K^6 for i16 points scaled by K: K <= 8 cannot overflow
static int check_k(uint32_t K, int32_t x0, int32_t y0, uint64_t w, uint64_t h)
{
    if (K < 1 || K > 8) return fail(FR_E_UNSUPPORTED, "K must be in [1, 8] (128-bit predicate range)");
    if (w > (1u << 20) || h > (1u << 20)) return fail(FR_E_UNSUPPORTED, "lattice larger than 2^20 per axis");
    const int64_t lim = (int64_t)1 << 20;           // query points stay within a few em of the scaled glyph
    if (x0 < -lim || x0 + (int64_t)w > lim || y0 > lim || y0 - (int64_t)h < -lim)
        return fail(FR_E_UNSUPPORTED, "lattice beyond +-2^20 scaled units");
    return FR_OK;
}

int fr_exact_lattice(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                     uint32_t n_contours, uint32_t K, int32_t x0, int32_t y0, uint32_t w, uint32_t h,
                     int16_t *out_host)
{
    int rc = check_k(K, x0, y0, w, h);
    if (rc) return rc;
    ExactDev d;
    rc = exact_setup(ctx, points_xy, contour_start, n_contours, d, (int)K);
    if (rc) return rc;
    return exact_run(ctx, d, nullptr, (uint64_t)w * h, w, x0, y0, out_host);
}

int fr_exact_coverage(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                      uint32_t n_contours, uint32_t K, int32_t x0, int32_t y0, uint32_t w_px, uint32_t h_px,
                      uint32_t n, uint8_t *out_host)
{
    if (n < 1 || n > 8) return fail(FR_E_INVALID, "samples per axis must be in [1, 8]");
    int rc = check_k(K, x0, y0, (uint64_t)w_px * n, (uint64_t)h_px * n);
    if (rc) return rc;
    ExactDev d;
    rc = exact_setup(ctx, points_xy, contour_start, n_contours, d, (int)K);
    if (rc) return rc;
    return exact_run(ctx, d, nullptr, (uint64_t)w_px * n * h_px * n, w_px * n, x0, y0, out_host, w_px, h_px, n);
}


int fr_glyph_debug_render(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                          uint32_t n_contours, const int16_t box[4], uint8_t winding_scale, uint8_t *rgb_host)
{
    if (!box) return fail(FR_E_INVALID, "box is NULL");
    if (!rgb_host) return fail(FR_E_INVALID, "rgb_host is NULL");
    const int W = (int)box[2] - box[0] + 3, H = (int)box[3] - box[1] + 3;       // Image.zig:183
    if (W < 1 || H < 1) return fail(FR_E_INVALID, "empty box");
    ExactDev d;
    int rc = exact_setup(ctx, points_xy, contour_start, n_contours, d);
    if (rc) return rc;
    const uint64_t nq = (uint64_t)W * H;
    int16_t *d_lat = nullptr;
    uint8_t *d_rgb = nullptr;
    hipError_t e = hipMalloc(&d_lat, nq * 2);
    if (e == hipSuccess) e = hipMalloc(&d_rgb, nq * 3);
    if (e == hipSuccess) {
        // the lattice GlyphDebug.render walks (Image.zig:227-236), coloured on the device (setWindingLinear)
        fr::launch_exact_winding(d.pts, d.seg_p0, d.ctype, d.inc, d.n_seg, nullptr, nq, (uint32_t)W, box[0] - 1, box[3] + 1, 1, d_lat, ctx->stream);
        fr::launch_glyph_debug_color(d_lat, nq, winding_scale, 150u, d_rgb, ctx->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(rgb_host, d_rgb, nq * 3, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = e2;
    dfree(d_lat); dfree(d_rgb);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? FR_E_NOMEM : FR_E_HIP, "fr_glyph_debug_render: %s", hipGetErrorString(e));
    // setGlyphPoints (Image.zig:202-218): a few writes in the reference's own order (later points overwrite
    // earlier ones where they coincide), done on the host on the image just copied back
    for (uint32_t c = 0; c < n_contours; ++c) {
        const int16_t *cp = points_xy + 2u * (size_t)contour_start[c];
        const uint32_t curves = (contour_start[c + 1] - contour_start[c]) / 2u;
        for (uint32_t k = 0; k < curves; ++k) {
            for (int which = 0; which < 2; ++which) {
                const int16_t *pt = cp + 2u * (2u * k + (uint32_t)which);
                const int64_t wq = (int64_t)pt[0] - box[0] + 1, hq = (int64_t)box[3] - pt[1] + 1;
                if (wq < 0 || hq < 0 || wq >= W || hq >= H)
                    return fail(FR_E_INVALID, "glyph point outside its box (the reference would write out of bounds)");
                uint8_t *px = rgb_host + 3u * ((size_t)hq * W + (size_t)wq);
                if (which == 0) { px[0] = 255; px[1] = 255; px[2] = 0; }        // on-curve
                else { px[0] = 0; px[1] = 255; px[2] = 255; }                   // control
            }
        }
    }
    return FR_OK;
}

int fr_atlas_layout(const int16_t *boxes, uint32_t n_glyphs, uint32_t first_glyph,
                    const uint16_t *units_per_em, uint32_t n_upm, uint16_t font_size,
                    uint32_t cell, uint32_t cols, uint32_t rows_per_page,
                    fr_job *jobs_out, uint32_t *page_of_job, uint32_t *n_pages)
{
    if (n_glyphs && (!boxes || !jobs_out)) return fail(FR_E_INVALID, "fr_atlas_layout: NULL argument");
    if (!units_per_em || (n_upm != 1 && n_upm != n_glyphs)) return fail(FR_E_INVALID, "units_per_em: one value or one per glyph");
    if (cell == 0 || cols == 0 || font_size == 0) return fail(FR_E_INVALID, "cell, cols and font_size must be > 0");
    if ((uint64_t)cell * cols > 0xffffffffull) return fail(FR_E_UNSUPPORTED, "atlas wider than 2^32 pixels");
    const uint64_t per_page = rows_per_page ? (uint64_t)rows_per_page * cols : 0;
    for (uint32_t i = 0; i < n_glyphs; ++i) {
        const uint16_t upm = units_per_em[n_upm == 1 ? 0 : i];
        if (upm == 0) return fail(FR_E_INVALID, "units_per_em must be > 0");
        const float scale = (float)font_size / (float)upm;                              // render_glyph.zig:13
        const float fx = std::floor((float)boxes[4 * (size_t)i + 0] * scale);           // :15-16
        const float fy = std::ceil((float)boxes[4 * (size_t)i + 3] * scale);            // :15, :17
        const uint64_t slot = per_page ? i % per_page : i;
        const uint64_t row = slot / cols;
        if (row * cell > 0xffffffffull) return fail(FR_E_UNSUPPORTED, "atlas taller than 2^32 pixels: use pages");
        fr_job &jb = jobs_out[i];
        jb.glyph = first_glyph + i;
        jb.min_x = (int32_t)fx; jb.max_y = (int32_t)fy;
        jb.w = cell; jb.h = cell;
        jb.out_x = (uint32_t)(slot % cols) * cell;
        jb.out_y = (uint32_t)row * cell;
        jb.scale = scale;
        if (page_of_job) page_of_job[i] = per_page ? (uint32_t)(i / per_page) : 0u;
    }
    if (n_pages) *n_pages = per_page ? (uint32_t)((n_glyphs + per_page - 1) / per_page) : (n_glyphs ? 1u : 0u);
    return FR_OK;
}

// renderGlyph's own image per glyph (render_glyph.zig:13-19 through fr_render_glyph_dims), shelf-packed in input order
int fr_atlas_layout_glyph_dims(const int16_t *boxes, uint32_t n_glyphs, uint32_t first_glyph,
                               const uint16_t *units_per_em, uint32_t n_upm, uint16_t font_size,
                               uint32_t atlas_w, uint32_t align, fr_job *jobs_out, uint32_t *atlas_h)
{
    if (n_glyphs && (!boxes || !jobs_out)) return fail(FR_E_INVALID, "fr_atlas_layout_glyph_dims: NULL argument");
    if (!units_per_em || (n_upm != 1 && n_upm != n_glyphs)) return fail(FR_E_INVALID, "units_per_em: one value or one per glyph");
    if (atlas_w == 0 || font_size == 0) return fail(FR_E_INVALID, "atlas_w and font_size must be > 0");
    if (align == 0) align = 1;
    uint64_t x = 0, y = 0, shelf_h = 0;
    for (uint32_t i = 0; i < n_glyphs; ++i) {
        int16_t mn[2], mx[2];
        uint16_t w, h;
        float scale;
        const int rc = fr_render_glyph_dims(boxes + 4 * (size_t)i, units_per_em[n_upm == 1 ? 0 : i], font_size, mn, mx, &w, &h, &scale);
        if (rc) return rc;
        if (w > atlas_w) return fail(FR_E_INVALID, "glyph %u is %u pixels wide, the atlas %u", i, (unsigned)w, atlas_w);
        x = (x + align - 1) / align * align;
        if (x + w > atlas_w) { x = 0; y += shelf_h; shelf_h = 0; }          // next shelf
        if (y + h > 0xffffffffull) return fail(FR_E_UNSUPPORTED, "atlas taller than 2^32 pixels");
        fr_job &jb = jobs_out[i];
        jb.glyph = first_glyph + i;
        jb.min_x = mn[0]; jb.max_y = mx[1];                                  // :26-27: sample (min_x + x, max_y - y)
        jb.w = w; jb.h = h;
        jb.out_x = (uint32_t)x; jb.out_y = (uint32_t)y;
        jb.scale = scale;
        x += w;
        shelf_h = std::max<uint64_t>(shelf_h, h);
    }
    if (atlas_h) *atlas_h = (uint32_t)std::min<uint64_t>(y + shelf_h, 0xffffffffull);
    return FR_OK;
}

}  // extern "C"
