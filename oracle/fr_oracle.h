/*
 * fr_oracle.h — CPU restatement of the reference's glyph winding / raster path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / reported baseline.  The shipped path is the HIP
 * library in font-renderer_amd/csrc (libfr_raster.so) and never calls in here.
 *
 * PARITY UNPINNED: the reference (nyasyamorina/font-renderer, Zig 0.15) ships no
 * tests, golden vectors or fixtures for this path and cannot be built in this
 * image (no zig toolchain).  Every function below is a restatement of the Zig
 * *source text*; each cites the file:line it follows (paths relative to
 * /root/reference).  Cross-checks available: SURVEY.md Appendix B (an
 * independent numpy emulation) and tests/ref_numpy.py (a second, independent
 * restatement used to cross-validate this one).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (strict IEEE-754 binary32,
 * no FMA contraction).
 */
#ifndef FR_ORACLE_H
#define FR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Glyph (src/font/Glyph.zig:11-24) flattened for a C caller:
 *   pts     : interleaved i16 (x,y) pairs, all contours back to back
 *             (Point(i16) is an extern struct {x,y}: src/tools/geometry.zig:7-11)
 *   cstart  : n_contours+1 offsets, in points, into pts
 *   contour k has points [cstart[k], cstart[k+1]); even = on-curve, odd = control,
 *   last == first (Glyph.zig:23).
 */

/* ---- a3: contour producer (Glyph.zig:26-74, geometry.zig:12-17) ---- */
uint32_t or_count_ttf_points(const uint16_t *end_pts, uint32_t n_contours,
                             const uint8_t *on_curve);
/* writes expanded contour `ci` into out (capacity >= its count); returns count */
uint32_t or_contour_init_ttf(const int16_t *coords_xy, const uint8_t *on_curve,
                             const uint16_t *end_pts, uint32_t ci, int16_t *out_xy);
/* composite part transform (Glyph.zig:178-182) + rounding (Glyph.zig:141-149,
 * helpers.zig:112-123). returns 0 ok, -1 if the reference would @panic("not impl") */
int or_transform_point(int16_t x, int16_t y, const int16_t m2d14[4], int16_t e, int16_t f,
                       int round_to_grid, int16_t *ox, int16_t *oy);

/* ---- a2: f32 ray-cast winding (render_glyph.zig:35-73) ---- */
int16_t or_glyph_winding_at(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                            float cx, float cy);

/* ---- a1: renderGlyph (render_glyph.zig:11-33) ---- */
void or_render_glyph_dims(const int16_t box[4], uint16_t units_per_em, uint16_t font_size,
                          int16_t min_corner[2], int16_t max_corner[2],
                          uint16_t *width, uint16_t *height, float *scale);
/* out: width*height u8, row-major, value clamp(w*20+100,0,255) */
void or_render_glyph(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                     const int16_t box[4], uint16_t units_per_em, uint16_t font_size,
                     uint8_t *out);

/* ---- generic cell renderer: the reference's sample formula on a caller-chosen
 * grid ("cell configs are build-defined fixed grids using the same
 * sample-point formula").  Sample (x,y,i,j) of an n x n sub-grid:
 *     px = f32(min_x + x) + off(i)      py = f32(max_y - y) - off(j)
 *     off(k) = (k + phase)/n, phase = 0 (corner; n=1 reproduces render_glyph.zig:26-27)
 *                                     or 0.5 (centre)
 *     cx = px / scale, cy = py / scale  (IEEE division, as render_glyph.zig:26-27)
 * modes (values shared with include/fr_raster.h):
 *   0 WINDING_I16  : int16 winding of sample (0,0)           (n must be 1)
 *   1 GRAY_DEBUG   : u8 clamp(w*20+100,0,255)                (render_glyph.zig:28)
 *   2 MASK_NONZERO : u8 (w != 0) ? 255 : 0                   (render_glyph.zig:29)
 *   3 COVERAGE_U8  : u8 round_half_up(255*k/(n*n)), k = #samples with w != 0
 *                    (non-zero rule :29, box filter = VulkanContext.zig:312 average resolve)
 *   4 SDF_U8       : BUILD-DEFINED signed distance field, twin of csrc/fr_sdf.hip (n must be 1):
 *                    u8 = clamp(floor(128 + 16*d + 0.5)), d in pixels, + where w != 0
 * out_stride in elements. */
enum { OR_WINDING_I16 = 0, OR_GRAY_DEBUG = 1, OR_MASK_NONZERO = 2, OR_COVERAGE_U8 = 3, OR_SDF_U8 = 4 };
int or_render_cell(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                   int32_t min_x, int32_t max_y, uint32_t w, uint32_t h, float scale,
                   int mode, int n, int phase_center, void *out, size_t out_stride);

/* batch form used by bench.py's cpu_baseline and by atlas parity tests.
 * glyph g uses contours [gstart[g], gstart[g+1]); job arrays are per glyph. */
typedef struct {
    uint32_t glyph;      /* index into gstart */
    int32_t min_x, max_y;
    uint32_t w, h;
    uint32_t out_x, out_y;
    float scale;
} or_job;
int or_render_batch(const int16_t *pts, const uint32_t *cstart, const uint32_t *gstart,
                    const or_job *jobs, uint32_t n_jobs, int mode, int n, int phase_center,
                    void *out, size_t out_stride, int n_threads);

/* ---- a4: GlyphInfo / CurveType (render_glyph.zig:76-155) ---- */
enum {
    OR_X_AXIS = 0, OR_BALANCE, OR_UP_STRIGHT, OR_UP_NORMAL, OR_UP_U, OR_UP_INV_U,
    OR_DOWN_STRIGHT, OR_DOWN_NORMAL, OR_DOWN_INV_U, OR_DOWN_U
};
uint8_t or_curve_type(const int16_t p0[2], const int16_t p1[2], const int16_t p2[2]);
/* one entry per curve, contours back to back: type[] and include_p0[] */
void or_glyph_info_init(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                        uint8_t *curve_type, uint8_t *include_p0);

/* ---- a5-a7: exact integer winding (render_glyph.zig:160-300) ---- */
int16_t or_winding_in_glyph(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                            const uint8_t *curve_type, const uint8_t *include_p0,
                            int16_t px, int16_t py);
int or_solve2_roots_winding(const int32_t p[2], const int32_t p0[2], const int32_t p1[2],
                            const int32_t p2[2]);
int or_solve1_root_crossing(const int32_t p[2], const int32_t p0[2], const int32_t p1[2],
                            const int32_t p2[2], int tilt_up);
/* diagnostics: how often the restatement left the reference's defined domain */
uint64_t or_diag_i64_overflow_count(void);  /* products that exceed i64 (reference: UB/trap) */
uint64_t or_diag_assert_fail_count(void);   /* render_glyph.zig:284 assert(dy >= 0) violated */
void or_diag_reset(void);

/* lattice dump used by Image.GlyphDebug.render (Image.zig:220-240):
 * out is (x_max-x_min+3) x (y_max-y_min+3) int16 windings, point (x_min+w-1, y_max-h+1) */
void or_winding_lattice(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                        const int16_t box[4], int16_t *out);
/* SURVEY §8 f-3, build-defined: a4-a7 on the glyph scaled by K at the integer points (x0+i, y0-j) */
void or_exact_lattice(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours, uint32_t K,
                      int32_t x0, int32_t y0, uint32_t w, uint32_t h, int16_t *out);
void or_exact_coverage(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours, uint32_t K,
                       int32_t x0, int32_t y0, uint32_t w_px, uint32_t h_px, uint32_t n, uint8_t *out);

/* ---- a8: Image sinks (Image.zig:44-130, 173-241) ---- */
void or_gray_rgb(uint8_t val, uint8_t rgb[3]);                                  /* :78-82 */
void or_winding_rgb(int16_t val, uint8_t scaler, uint8_t overflow_color, uint8_t rgb[3]); /* :121-129 */
/* GlyphDebug.render -> RGB image, 3 bytes per pixel (Image.zig:220-240) */
void or_glyph_debug_render(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                           const int16_t box[4], uint8_t winding_scale, uint8_t *rgb_out);

/* ---- f2: QOI RGB writer (src/tools/qoi.zig:25-88). returns bytes written
 * (<= cap) or 0 if cap too small. rgb = w*h*3 bytes. */
size_t or_qoi_encode_rgb(const uint8_t *rgb, uint32_t w, uint32_t h, uint8_t *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
