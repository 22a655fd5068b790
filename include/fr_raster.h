/*
 * fr_raster.h — C ABI of the MI355X-native glyph rasterizer (libfr_raster.so).
 *
 * This is the drop-in boundary for ONE path of nyasyamorina/font-renderer: the
 * per-pixel winding / coverage loop of src/tools/render_glyph.zig (and the
 * TriangulatedGlyph + shader.slang + Vulkan pipeline that draws the same filled
 * region).  The host keeps the reference's Glyph (src/font/Glyph.zig:11-24) and
 * Image (src/tools/Image.zig:44-130) types; it hands this library flat views of
 * the glyph's contiguous i16 points and a caller-allocated output buffer.
 * Reference-side binding: see INTEGRATION.md and bindings/fr_raster.zig.
 *
 * Plain pointers and sizes only; no C++ / torch types.  All functions return an
 * fr_status (0 = ok, < 0 = error) unless noted; fr_last_error() returns a
 * thread-local message for the last failing call.  The library never frees or
 * retains host pointers past a call, never aborts and never throws across the
 * ABI.  One fr_ctx is used from one thread at a time (the reference is
 * single-threaded: SURVEY §8b).
 *
 * There is NO CPU fallback: every compute entry point runs hand-written HIP
 * kernels on a gfx950 device and fails with FR_E_HIP when none is usable.
 */
#ifndef FR_RASTER_H
#define FR_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FR_ABI_VERSION 1

typedef enum fr_status {
    FR_OK = 0,
    FR_E_INVALID = -1,      /* bad argument / malformed glyph tables               */
    FR_E_HIP = -2,          /* HIP runtime error or no usable device               */
    FR_E_NOMEM = -3,        /* host or device allocation failed                    */
    FR_E_UNSUPPORTED = -4   /* valid request outside the implemented envelope      */
} fr_status;

/* What one output element holds.  The value maps are the reference's own:
 *   FR_WINDING_I16  int16 winding number at the sample      (Image.Winding.data, Image.zig:85-88;
 *                                                            glyphWindingAt, render_glyph.zig:35-73)
 *   FR_GRAY_DEBUG   u8 clamp(w*20+100, 0, 255)               (render_glyph.zig:28 — what renderGlyph emits)
 *   FR_MASK_NONZERO u8 (w != 0) ? 255 : 0                    (render_glyph.zig:29, the commented alternative)
 *   FR_COVERAGE_U8  u8 round(255*k/n^2), k = # of the n x n sub-samples with w != 0
 *                   (non-zero fill as :29; box filter = the MSAA average resolve,
 *                    VulkanContext.zig:307-313 — the reference's only anti-aliasing)        */
typedef enum fr_mode {
    FR_WINDING_I16 = 0,
    FR_GRAY_DEBUG = 1,
    FR_MASK_NONZERO = 2,
    FR_COVERAGE_U8 = 3,
    FR_SDF_U8 = 4           /* BUILD-DEFINED (the reference has no SDF): u8 = clamp(round(128 + 16*d)),
                               d = distance in pixels from the sample to the nearest quadratic segment,
                               + inside (winding != 0) / - outside; n must be 1.  Exact definition:
                               font-renderer_amd/csrc/fr_sdf.hip                                         */
} fr_mode;

/* Sub-sample k of an axis sits at (k + phase)/n of a pixel:
 *   FR_SAMPLE_CORNER phase 0   — with n = 1 exactly the reference's sample, the pixel
 *                                *corner* (min_x + x, max_y - y) (render_glyph.zig:26-27)
 *   FR_SAMPLE_CENTER phase 1/2 — regular n x n grid centred in the pixel              */
typedef enum fr_sample_phase { FR_SAMPLE_CORNER = 0, FR_SAMPLE_CENTER = 1 } fr_sample_phase;

typedef struct fr_raster_params {
    int32_t mode;              /* fr_mode                                              */
    int32_t samples_per_axis;  /* n in {1,2,4}; must be 1 unless mode = FR_COVERAGE_U8  */
    int32_t sample_phase;      /* fr_sample_phase                                      */
    int32_t reserved;          /* 0                                                    */
} fr_raster_params;

/* One glyph cell to rasterize.  Sample point of pixel (x, y), sub-sample (i, j):
 *     cx = (f32(min_x + x) + (i + phase)/n) / scale
 *     cy = (f32(max_y - y) - (j + phase)/n) / scale        (render_glyph.zig:26-27: division,
 *                                                            y-down image, font-unit ray origin)
 * renderGlyph's own grid is min = floor(box_min*scale), max = ceil(box_max*scale),
 * w = max_x-min_x+1, h = max_y-min_y+1 (render_glyph.zig:13-19): fr_render_glyph_dims. */
typedef struct fr_job {
    uint32_t glyph;            /* index into the glyph set                             */
    int32_t  min_x, max_y;     /* pixel coordinate of column 0 / row 0                 */
    uint32_t w, h;             /* cell size in pixels                                  */
    uint32_t out_x, out_y;     /* destination of the cell's (0,0) in the output, elements / rows */
    float    scale;            /* font_size / units_per_em (render_glyph.zig:13)       */
} fr_job;

typedef struct fr_ctx fr_ctx;
typedef struct fr_glyphset fr_glyphset;
typedef struct fr_plan fr_plan;

/* ---- library / context -------------------------------------------------- */
int fr_abi_version(void);
const char *fr_last_error(void);
/* a string that changes whenever a source file of the library does: "r03-" + the first 12 hex digits of the SHA-256 of
 * csrc's sources, computed by the Makefile at build time (profiles/traffic.json is keyed by it) */
const char *fr_build_id(void);

/* device: HIP device ordinal.  hip_stream: a hipStream_t to launch on (e.g. the
 * caller's torch stream), or NULL to let the context create and own one.       */
int fr_ctx_create(int device, void *hip_stream, fr_ctx **out);
void fr_ctx_destroy(fr_ctx *ctx);
int fr_ctx_sync(fr_ctx *ctx);
/* tuning / test knobs: "kmax" (crossings kept per sample row, in registers, before the
 * exact direct-sum fallback: rounded up to 8, 16 or 32; default 32), "strip_px" (column
 * strip width in pixels, multiple of 16, <= 256: wider cells are rendered strip by strip; the fast kernels use
 * the largest of 64 / 128 / 256 that does not exceed it),
 * "cov4" (0: every job takes the general kernel), "sdf_cull" (0: FR_SDF_U8 looks at every segment from
 * every pixel — the culls are exact, this is how the tests show it), "zero_copy" (1: fr_render_glyph renders small
 * glyphs straight from / into pinned host memory; measured no faster, off by default), "overlap" (a plan that needs several kernel launches forks the
 * smaller ones onto an internal second stream and joins them: 1 (default) for plans of >= 32 Mpixel — below that one
 * stream is quicker —, 2 always, 0 never), "graph" (1: the launches of a plan's render — fork and join included — are
 * captured into a hipGraph at the first fr_plan_render to a destination and replayed with one hipGraphLaunch afterwards;
 * captured again when the destination or an option changes.  Measured: worth 3 - 6 % on a plan of four
 * launches forked over two streams, a loss of ~5 us per render on plans of one to three launches; default 0), "min_wgs", "fuse_prepare", "lds_pad"   */
int fr_ctx_set_option(fr_ctx *ctx, const char *key, int64_t value);

/* ---- glyph sets: Glyph[] flattened (Glyph.zig:11-24) ---------------------
 * points_xy      : i16 (x,y) pairs, every contour of every glyph back to back — the
 *                  reference already keeps one glyph's points in one allocation
 *                  (Glyph.zig:89-96), so a single glyph is passed without copying;
 * contour_start  : n_contours+1 offsets (in points); contour c = [start[c], start[c+1]),
 *                  even index on-curve, odd index control, last == first (Glyph.zig:23),
 *                  so its length is odd and >= 3 (or 1: a degenerate contour, 0 curves);
 * glyph_start    : n_glyphs+1 offsets (in contours).
 * Uploads the points to HBM and runs the per-segment precompute kernel once
 * (root records with exact acceptance intervals).  Scale-independent:
 * one glyph set serves every font size, mode and sample count.                     */
int fr_glyphset_create(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                       uint32_t n_contours, const uint32_t *glyph_start, uint32_t n_glyphs,
                       fr_glyphset **out);
void fr_glyphset_destroy(fr_glyphset *gs);
/* re-runs the precompute kernel on the context's stream (asynchronous); for timing */
int fr_glyphset_prepare(fr_glyphset *gs);
/* totals, for reporting: segments (curves) and surviving root records */
int fr_glyphset_stats(const fr_glyphset *gs, uint64_t *n_segments, uint64_t *n_records);

/* ---- batched rasterization ----------------------------------------------
 * A plan keeps the job table resident on the device so a batch can be re-rendered
 * without host traffic (atlas pages, benchmark steps).                              */
int fr_plan_create(fr_ctx *ctx, const fr_glyphset *gs, const fr_job *jobs, uint32_t n_jobs,
                   const fr_raster_params *params, fr_plan **out);
void fr_plan_destroy(fr_plan *plan);
/* Every render starts from the glyph POINTS (nothing derived is reused between renders): the render
 * kernels build the root records of a glyph of <= 128 (general kernel) / <= 768 (cov4 / win1 kernels) segments
 * themselves, in LDS ("fused", decided per job); the precompute kernel is re-run first for the larger glyphs only.
 * Asynchronous on the context's stream.  out_dev: DEVICE pointer to an array of
 * out_rows rows of out_stride elements (u8, or i16 for FR_WINDING_I16); every job
 * must fit inside it (checked); out_stride <= 2^26 elements (FR_E_INVALID beyond: the kernels address the rows of a
 * wave band by 32-bit offsets from the band's base).  Pixels outside all jobs are not touched.                     */
int fr_plan_render(fr_plan *plan, void *out_dev, size_t out_stride, size_t out_rows);
/* same, bracketed by HIP events on the launch stream; synchronous; *ms = kernel time */
int fr_plan_render_timed(fr_plan *plan, void *out_dev, size_t out_stride, size_t out_rows, float *ms);
uint64_t fr_plan_pixels(const fr_plan *plan);   /* sum of w*h over the jobs */
/* how the plan's jobs are split between the render kernels (the decision is per job): the fast kernels — cov4_kernel
 * (4 x 4 or 2 x 2 samples) / win1_kernel (one sample per pixel) — take cells of ANY width and height up to 2048 sample
 * rows (renderGlyph's own image sizes, render_glyph.zig:14-19, included: strips of 64 / 128 / 256 pixels chosen from
 * the job's width, stores clipped at the cell's border) of glyphs with <= 768 segments; the general render_kernel takes
 * everything else                                                                                              */
int fr_plan_stats(const fr_plan *plan, uint32_t *n_jobs_cov4, uint32_t *n_jobs_general);
/* the kernel instances one render of the plan launches, as rocprofv3 --kernel-trace names them, each with its job
 * count: "fr::cov4_kernel<4, 32, 4, 4> x20992; fr::render_kernel<3, 4, 32, -1> x3" (NUL-terminated, truncated to cap) */
int fr_plan_describe(const fr_plan *plan, char *buf, size_t cap);

/* One-shot: plan + render + copy back.  out_host: HOST buffer (caller-allocated,
 * e.g. Image.Gray.data / Image.Winding.data from the Zig allocator).  Synchronous.  */
int fr_render_batch(fr_ctx *ctx, const fr_glyphset *gs, const fr_job *jobs, uint32_t n_jobs,
                    const fr_raster_params *params, void *out_host, size_t out_stride, size_t out_rows);

/* ---- renderGlyph drop-in (render_glyph.zig:11-33) -------------------------
 * fr_render_glyph_dims reproduces :13-19 on the host so the caller can size the
 * Image.Gray first; fr_render_glyph fills it: w*h u8, row-major, value per `mode`
 * (FR_GRAY_DEBUG is what the reference's renderGlyph returns).                      */
int fr_render_glyph_dims(const int16_t box[4], uint16_t units_per_em, uint16_t font_size,
                         int16_t min_corner[2], int16_t max_corner[2],
                         uint16_t *width, uint16_t *height, float *scale);
int fr_render_glyph(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                    uint32_t n_contours, const int16_t box[4], uint16_t units_per_em,
                    uint16_t font_size, int32_t mode, void *out_host);

/* ---- exact-integer path (render_glyph.zig:76-300) -------------------------
 * fr_glyph_info_init: GlyphInfo.init (:110-146) — one CurveType (:84-95 enum order)
 * and one include_p0 flag per curve, contours back to back, computed on the device.
 * fr_winding_in_glyph: windingInGlyph (:160-247) at n_query integer font-unit points.
 * fr_winding_lattice: the lattice Image.GlyphDebug.render walks (Image.zig:227-236):
 * (x_max-x_min+3) x (y_max-y_min+3) int16, point (x_min+w-1, y_max-h+1).
 * Predicates use 128-bit integers: identical to the reference's i64 wherever that
 * does not overflow.                                                  */
int fr_glyph_info_init(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                       uint32_t n_contours, uint8_t *curve_type, uint8_t *include_p0);
int fr_winding_in_glyph(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                        uint32_t n_contours, const int16_t *query_xy, uint32_t n_query,
                        int16_t *out_winding);
int fr_winding_lattice(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                       uint32_t n_contours, const int16_t box[4], int16_t *out_host);

/* Image.GlyphDebug.render (Image.zig:220-240): the lattice of fr_winding_lattice coloured by
 * setWindingLinear (:192-200; overflow colour 150) with the glyph's on-curve / control points marked
 * {255,255,0} / {0,255,255} by setGlyphPoints (:202-218, in the reference's order).  rgb_host receives
 * (x_max-x_min+3) * (y_max-y_min+3) RGB triples, row-major — Image.RGB.data (Image.zig:132-170).        */
int fr_glyph_debug_render(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                          uint32_t n_contours, const int16_t box[4], uint8_t winding_scale, uint8_t *rgb_host);

/* ---- exact-integer sampling on a K-times refined lattice (BUILD-DEFINED) ----
 * GlyphInfo.init + windingInGlyph (render_glyph.zig:110-146, :160-300) applied, rule for rule, to
 * the glyph whose points are multiplied by K (1 <= K <= 8, so the 128-bit predicates cannot
 * overflow), at the integer points (x0 + i, y0 - j), i < w, j < h of that scaled glyph — i.e. at
 * the font-unit points ((x0+i)/K, (y0-j)/K) with no floating point anywhere.  K = 1 is the
 * reference's own lattice.  fr_exact_coverage evaluates n x n lattice points per pixel
 * (lattice (w_px*n) x (h_px*n)) and writes round_half_up(255 * #{winding != 0} / n^2), the same
 * box-filtered non-zero fill as FR_COVERAGE_U8 on the integer inside test.  The reference has no
 * such mode (its exact path is dead code used only by GlyphDebug at K = 1); parity is against the
 * oracle's twin of the same definition.                                                        */
int fr_exact_lattice(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                     uint32_t n_contours, uint32_t K, int32_t x0, int32_t y0, uint32_t w, uint32_t h,
                     int16_t *out_host /* [h][w] */);
int fr_exact_coverage(fr_ctx *ctx, const int16_t *points_xy, const uint32_t *contour_start,
                      uint32_t n_contours, uint32_t K, int32_t x0, int32_t y0, uint32_t w_px, uint32_t h_px,
                      uint32_t n, uint8_t *out_host /* [h_px][w_px] */);

/* ---- atlas layout (host side; BUILD-DEFINED: the reference has no atlas) -------------------
 * The fixed cell grid every atlas of this library uses: glyph i of `n_glyphs` (index first_glyph + i in
 * the glyph set, box boxes[4 i ..]) gets a cell x cell window whose pixel (0,0) is the glyph's own
 * renderGlyph origin — scale = f32(font_size) / f32(units_per_em), min_x = floor(x_min * scale),
 * max_y = ceil(y_max * scale) in binary32 exactly as render_glyph.zig:13-17 — at column (i % cols),
 * row (i / cols) of a cols-wide grid.  rows_per_page > 0 cuts the grid into pages of that many cell rows
 * (a 2048^2 page of 128-pixel cells: cols = rows_per_page = 16): out_y restarts on every page and
 * page_of_job[i] (if not NULL) names the page; *n_pages (if not NULL) receives the page count.
 * units_per_em: n_upm = 1 (one font) or n_upm = n_glyphs (one value per glyph).                      */
int fr_atlas_layout(const int16_t *boxes, uint32_t n_glyphs, uint32_t first_glyph,
                    const uint16_t *units_per_em, uint32_t n_upm, uint16_t font_size,
                    uint32_t cell, uint32_t cols, uint32_t rows_per_page,
                    fr_job *jobs_out, uint32_t *page_of_job, uint32_t *n_pages);

/* The reference's own product shape in batch: glyph i gets exactly the image renderGlyph would allocate for it —
 * fr_render_glyph_dims' W x H, pixel (0,0) at (min_x, max_y) (render_glyph.zig:13-19, :26-27) — shelf-packed in input
 * order into an atlas `atlas_w` elements wide: left to right, out_x rounded up to a multiple of `align` (1: tight),
 * a new shelf when the next image does not fit; *atlas_h (if not NULL) receives the rows used.  Every job's bytes are
 * what fr_render_glyph writes for that glyph (tests).                                                          */
int fr_atlas_layout_glyph_dims(const int16_t *boxes, uint32_t n_glyphs, uint32_t first_glyph,
                               const uint16_t *units_per_em, uint32_t n_upm, uint16_t font_size,
                               uint32_t atlas_w, uint32_t align, fr_job *jobs_out, uint32_t *atlas_h);

/* ---- multi-GPU assembly (optional; SURVEY section 8e: "optional final assembly: all-gather of row bands") --------
 * The hot path needs no collective: every rank renders its own glyph range into its own band.  When one rank (or all)
 * wants the whole atlas, the bands are equal-sized slices of one buffer — rank r rendered into
 * atlas_dev + r * band_bytes — and this call runs RCCL's in-place ncclAllGather over them on the context's stream.
 * nccl_comm is an ncclComm_t the HOST created (ncclCommInitRank / torch.distributed own the rendezvous; the library
 * never does); the call binds to the RCCL that is already loaded in the process (dlsym, no link-time dependency) and
 * returns FR_E_UNSUPPORTED if there is none.  Asynchronous like fr_plan_render: fr_ctx_sync / stream order apply.   */
int fr_allgather_bands(fr_ctx *ctx, void *nccl_comm, void *atlas_dev, size_t band_bytes);
/* The same assembly onto ONE rank: rank `root` receives every other rank's band into that rank's slot of ITS atlas_dev
 * (one RCCL group of ncclRecv's), every other rank sends its own band (atlas_dev + rank * band_bytes) and receives
 * nothing — each peer's bytes cross its own xGMI link to the root once, instead of every rank ingesting the whole atlas.
 * root < 0 is fr_allgather_bands.  Asynchronous on the context's stream.                                              */
int fr_gather_bands(fr_ctx *ctx, void *nccl_comm, void *atlas_dev, size_t band_bytes, int root);

/* ---- contour producer (host side): TrueType glyf/loca -> Glyph contour layout ------------
 * What font/Font.zig + font/ttf.zig + font/Glyph.zig do in the reference (Font.initTTF :31,
 * loadGlyph :171, SimpleGlyph.initFromReader ttf.zig:759, ComponentGlyph ttf.zig:830,
 * Contour.initTTF Glyph.zig:43, initTTFComponent :108, transform1 :178), restated in C++ so
 * whole fonts can be batch-fed to fr_glyphset_create without the Zig host.  The font bytes are
 * copied; glyphs are parsed lazily and cached like Font.glyphs.  Where the reference
 * @panic("not impl")s these return FR_E_UNSUPPORTED (hinted glyphs unless FR_FONT_ALLOW_HINTED).
 * fr_font_glyph_measure gives the sizes to allocate, fr_font_glyph_fill writes the i16 (x,y)
 * points and the n_contours+1 contour offsets (relative to the glyph, starting at 0).           */
typedef struct fr_font fr_font;
#define FR_FONT_ALLOW_HINTED 1u
int fr_font_open(const void *ttf_bytes, size_t len, uint32_t flags, fr_font **out);
void fr_font_close(fr_font *font);
int fr_font_info(const fr_font *font, uint16_t *units_per_em, uint16_t *num_glyphs, int *y0_baseline);
int fr_font_char_to_glyph(const fr_font *font, uint32_t codepoint, uint16_t *glyph_index);
/* Font.getGlyph's advance_width (Font.zig:161-169, hmtx read as Font.zig:123-139 reads it), font units */
int fr_font_glyph_advance(const fr_font *font, uint16_t glyph_index, int16_t *advance_width);
int fr_font_glyph_measure(fr_font *font, uint16_t glyph_index, uint32_t *n_contours, uint32_t *n_points, int16_t box[4]);
int fr_font_glyph_fill(fr_font *font, uint16_t glyph_index, int16_t *points_xy, uint32_t *contour_start);

/* ---- QOI writer (host side), byte-compatible with tools/qoi.zig:25-88 saveRGB -----------
 * RGB-only stream, op order RUN -> INDEX -> DIFF -> LUMA -> RGB, run cap 62, BE header, 8-byte
 * trailer.  fr_qoi_encode_gray feeds an Image.Gray (or an atlas page) through getRGBLinear =
 * {v,v,v} (Image.zig:78-82).  *n_out receives the stream length; FR_E_INVALID if cap is short. */
size_t fr_qoi_bound(uint32_t width, uint32_t height);
int fr_qoi_encode_rgb(const uint8_t *rgb, uint32_t width, uint32_t height, uint8_t *out, size_t cap, size_t *n_out);
int fr_qoi_encode_gray(const uint8_t *gray, uint32_t width, uint32_t height, size_t stride, uint8_t *out, size_t cap, size_t *n_out);

/* ---- self-test: exhaustive device-side check of an arithmetic shortcut ----------
 * The render kernel computes t = num / d (render_glyph.zig:51,60-61; d an integer, |d| <= 2^17)
 * as a reciprocal multiply + FMA correction.  This compares that sequence with IEEE division
 * for EVERY binary32 significand and every integer divisor in [d_lo, d_hi] (both signs, three
 * binades) on the current HIP device; *mismatches must come back 0 for the shortcut to be
 * admissible.  The full range 1..131072 takes a few seconds.                              */
int fr_selftest_division(uint32_t d_lo, uint32_t d_hi, uint64_t *mismatches,
                         uint32_t *bad_divisor, uint32_t *bad_x_bits);
/* The render kernel's square root (delta of render_glyph.zig:60) skips the denormal pre-scaling of
 * the general lowering; this compares it with the correctly rounded sqrt for every binary32 in
 * [2^-30, 2^66) on the device.  *mismatches must come back 0.                                  */
int fr_selftest_sqrt(uint64_t *mismatches, uint32_t *bad_x_bits);

#ifdef __cplusplus
}
#endif
#endif /* FR_RASTER_H */
