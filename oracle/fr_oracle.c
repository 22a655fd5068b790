/*
 * fr_oracle.c — CPU restatement of the reference's glyph winding / raster path.
 *
 * TEST INFRASTRUCTURE ONLY (see fr_oracle.h).  PARITY UNPINNED: restated from the
 * Zig source text; the reference holds no golden vectors and cannot be built here.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference).  Written to be read next to the Zig:
 * brute force, one statement per reference statement, no reformulation.
 *
 * gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread
 */
#include "fr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* a3: contour producer                                                       */
/* ------------------------------------------------------------------------- */

/* geometry.zig:12-17  Point.initMiddle: @divTrunc(a+b, 2) — C '/' truncates too.
 * (reference sums in i16; the restatement sums in int: same value wherever the
 * reference does not overflow, SURVEY §7 "i16 overflow domain") */
static void init_middle(const int16_t a[2], const int16_t b[2], int16_t out[2])
{
    out[0] = (int16_t)(((int)a[0] + (int)b[0]) / 2);
    out[1] = (int16_t)(((int)a[1] + (int)b[1]) / 2);
}

/* Glyph.zig:26-41  Contour.countTTFPoints */
uint32_t or_count_ttf_points(const uint16_t *end_pts, uint32_t n_contours, const uint8_t *on_curve)
{
    uint32_t count = n_contours;                      /* :27 */
    uint32_t coord_start = 0;                         /* :29 */
    for (uint32_t c = 0; c < n_contours; ++c) {       /* :30 */
        uint32_t coord_end = end_pts[c];
        int prev_on = on_curve[coord_end] != 0;       /* :31 */
        for (uint32_t i = coord_start; i < coord_end + 1; ++i) { /* :32 */
            int curr_on = on_curve[i] != 0;
            count += (prev_on == curr_on) ? 2 : 1;    /* :34 */
            prev_on = curr_on;
        }
        coord_start = coord_end + 1;                  /* :37 */
    }
    return count;
}

/* Glyph.zig:43-74  Contour.initTTF */
uint32_t or_contour_init_ttf(const int16_t *coords_xy, const uint8_t *on_curve,
                             const uint16_t *end_pts, uint32_t ci, int16_t *out_xy)
{
    uint32_t coord_start = (ci == 0) ? 0 : (uint32_t)end_pts[ci - 1] + 1;   /* :44 */
    uint32_t coord_end = end_pts[ci];                                       /* :45 */

    int prev_on = on_curve[coord_end] != 0;                                 /* :47 */
    int16_t prev_coord[2] = { coords_xy[2 * coord_end], coords_xy[2 * coord_end + 1] }; /* :48 */
    uint32_t next_point = prev_on ? 1 : 0;                                  /* :49 */
    for (uint32_t i = coord_start; i < coord_end + 1; ++i) {                /* :50 */
        int curr_on = on_curve[i] != 0;
        int16_t curr_coord[2] = { coords_xy[2 * i], coords_xy[2 * i + 1] };
        if (prev_on == curr_on) {                                           /* :54 */
            init_middle(prev_coord, curr_coord, &out_xy[2 * next_point]);   /* :57 */
            next_point += 1;
        }
        out_xy[2 * next_point] = curr_coord[0];                             /* :60 */
        out_xy[2 * next_point + 1] = curr_coord[1];
        next_point += 1;
        prev_on = curr_on;                                                  /* :63 */
        prev_coord[0] = curr_coord[0];
        prev_coord[1] = curr_coord[1];
    }
    if (prev_on) {                                                          /* :67 */
        out_xy[0] = out_xy[2 * (next_point - 1)];
        out_xy[1] = out_xy[2 * (next_point - 1) + 1];
    } else {                                                                /* :69 */
        out_xy[2 * next_point] = out_xy[0];
        out_xy[2 * next_point + 1] = out_xy[1];
        next_point += 1;
    }
    return next_point;                                                      /* :73 */
}

/* Glyph.zig:178-182 transform1 (i2d14 matrix entries, i18d14 result) and the
 * rounding at Glyph.zig:141-149 (FixedPointNumber.roundToInt/toInt, helpers.zig:103-123) */
static int32_t transform1(int16_t x, int16_t y, int16_t a_or_b, int16_t c_or_d, int16_t e_or_f)
{
    int32_t aa = a_or_b < 0 ? -(int32_t)a_or_b : a_or_b;   /* @abs -> u16 */
    int32_t cc = c_or_d < 0 ? -(int32_t)c_or_d : c_or_d;
    int32_t tmp = aa > cc ? aa : cc;                        /* :179 */
    /* :180  @abs(@as(i16, @bitCast(@abs(a) -% @abs(c)))) <= 8 */
    uint16_t diff_u = (uint16_t)((uint16_t)aa - (uint16_t)cc);
    int16_t diff_i = (int16_t)diff_u;
    int32_t diff_abs = diff_i < 0 ? -(int32_t)diff_i : diff_i;
    int sh = diff_abs <= 8 ? 1 : 0;
    int16_t shift = (int16_t)((int32_t)e_or_f << sh);       /* i16 shift-left, :180 */
    return (int32_t)a_or_b * (int32_t)x + (int32_t)c_or_d * (int32_t)y + tmp * (int32_t)shift; /* :181 */
}

static int32_t fx14_round(int32_t d)   /* helpers.zig:112-123 with bias_bits = 14 */
{
    int32_t base = d >> 14;
    if (d < 0) {
        if (d == INT32_MIN) return base;
        return ((-d) & (1 << 13)) == 0 ? base : base - 1;
    }
    return (d & (1 << 13)) == 0 ? base : base + 1;
}

int or_transform_point(int16_t x, int16_t y, const int16_t m[4], int16_t e, int16_t f,
                       int round_to_grid, int16_t *ox, int16_t *oy)
{
    int32_t fx = transform1(x, y, m[0], m[2], e);   /* :138 */
    int32_t fy = transform1(x, y, m[1], m[3], f);   /* :139 */
    if (round_to_grid) {                            /* :141-142 */
        *ox = (int16_t)fx14_round(fx);
        *oy = (int16_t)fx14_round(fy);
        return 0;
    }
    int32_t xg = fx >> 14, yg = fy >> 14;           /* :144-145 toInt */
    if ((xg << 14) != fx || (yg << 14) != fy) return -1;   /* :146-147 @panic("not impl") */
    *ox = (int16_t)xg;
    *oy = (int16_t)yg;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a2: glyphWindingAt  (render_glyph.zig:35-73)                                */
/* ------------------------------------------------------------------------- */
int16_t or_glyph_winding_at(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                            float cx, float cy)
{
    int16_t winding = 0;                                                /* :36 */
    for (uint32_t c = 0; c < n_contours; ++c) {                         /* :37 */
        const int16_t *cp = pts + 2 * (size_t)cstart[c];
        uint32_t len = cstart[c + 1] - cstart[c];
        uint32_t curve_count = len / 2;                                 /* :38 */
        for (uint32_t k = 0; k < curve_count; ++k) {                    /* :39 */
            /* :40-45  i16 -> f32 */
            const float p0x = (float)cp[2 * (2 * k) + 0],     p0y = (float)cp[2 * (2 * k) + 1];
            const float p1x = (float)cp[2 * (2 * k + 1) + 0], p1y = (float)cp[2 * (2 * k + 1) + 1];
            const float p2x = (float)cp[2 * (2 * k + 2) + 0], p2y = (float)cp[2 * (2 * k + 2) + 1];

            float a = p0y - 2 * p1y + p2y;                              /* :48 */
            if (a == 0) {                                               /* :49 */
                if (p2y == p0y) continue;                               /* :50 */
                float t = (cy - p0y) / (p2y - p0y);                     /* :51 */
                if (t < 0 || t >= 1) continue;                          /* :52 */
                float xx = ((p0x - 2 * p1x + p2x) * t + 2 * (p1x - p0x)) * t + p0x; /* :53 */
                if (xx < cx) continue;                                  /* :54 */
                winding += (p0y < p2y) ? -1 : 1;                        /* :55 */
                continue;                                               /* :56 */
            }
            float delta = cy * a + p1y * p1y - p0y * p2y;               /* :58 */
            if (delta < 0) continue;                                    /* :59 */
            float t_pos = ((p0y - p1y) + sqrtf(delta)) / a;             /* :60 */
            float t_neg = ((p0y - p1y) - sqrtf(delta)) / a;             /* :61 */
            float ts[2] = { t_pos, t_neg };
            for (int r = 0; r < 2; ++r) {                               /* :63 */
                float t = ts[r];
                if (t < 0 || t >= 1) continue;                          /* :64 */
                float xx = ((p0x - 2 * p1x + p2x) * t + 2 * (p1x - p0x)) * t + p0x; /* :65 */
                if (xx < cx) continue;                                  /* :66 */
                float dy = a * t + (p1y - p0y);                         /* :67 */
                winding += (dy > 0) ? -1 : 1;                           /* :68 */
            }
        }
    }
    return winding;                                                     /* :72 */
}

/* ------------------------------------------------------------------------- */
/* a1: renderGlyph  (render_glyph.zig:11-33)                                   */
/* ------------------------------------------------------------------------- */
void or_render_glyph_dims(const int16_t box[4], uint16_t units_per_em, uint16_t font_size,
                          int16_t min_corner[2], int16_t max_corner[2],
                          uint16_t *width, uint16_t *height, float *scale_out)
{
    float scale = (float)font_size / (float)units_per_em;              /* :13 */
    float b0 = (float)box[0] * scale, b1 = (float)box[1] * scale;      /* :14-15 */
    float b2 = (float)box[2] * scale, b3 = (float)box[3] * scale;
    min_corner[0] = (int16_t)floorf(b0);                               /* :16 */
    min_corner[1] = (int16_t)floorf(b1);
    max_corner[0] = (int16_t)ceilf(b2);                                /* :17 */
    max_corner[1] = (int16_t)ceilf(b3);
    *width = (uint16_t)(max_corner[0] - min_corner[0] + 1);            /* :18 */
    *height = (uint16_t)(max_corner[1] - min_corner[1] + 1);           /* :19 */
    if (scale_out) *scale_out = scale;
}

static uint8_t gray_debug_value(int16_t w)
{
    int v = (int)w * 20 + 100;                                         /* :28 */
    if (v < 0) v = 0;
    if (v > 255) v = 255;
    return (uint8_t)v;
}

void or_render_glyph(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                     const int16_t box[4], uint16_t units_per_em, uint16_t font_size, uint8_t *out)
{
    int16_t mn[2], mx[2];
    uint16_t width, height;
    float scale;
    or_render_glyph_dims(box, units_per_em, font_size, mn, mx, &width, &height, &scale);
    for (uint32_t y = 0; y < height; ++y) {                            /* :24 */
        for (uint32_t x = 0; x < width; ++x) {                         /* :25 */
            float coord_x = (float)(int16_t)(mn[0] + (int16_t)x) / scale;  /* :26 */
            float coord_y = (float)(int16_t)(mx[1] - (int16_t)y) / scale;  /* :27 */
            out[(size_t)y * width + x] =
                gray_debug_value(or_glyph_winding_at(pts, cstart, n_contours, coord_x, coord_y)); /* :28 */
        }
    }
}

/* ------------------------------------------------------------------------- */
/* generic cell renderer (build-defined products on the reference's sample formula) */
/* ------------------------------------------------------------------------- */
static uint8_t sdf_value(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours, float qx, float qy, float scale);

static float sub_offset(int k, int n, int phase_center)
{
    /* (k + phase)/n, exact in binary32 for n in {1,2,4,8,16} */
    return ((float)k + (phase_center ? 0.5f : 0.0f)) / (float)n;
}

int or_render_cell(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                   int32_t min_x, int32_t max_y, uint32_t w, uint32_t h, float scale,
                   int mode, int n, int phase_center, void *out, size_t out_stride)
{
    if (n != 1 && n != 2 && n != 4 && n != 8 && n != 16) return -1;
    if ((mode == OR_WINDING_I16 || mode == OR_SDF_U8) && n != 1) return -1;
    if (mode < 0 || mode > OR_SDF_U8) return -1;
    if (mode == OR_SDF_U8) {
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) {
                const float off = phase_center ? 0.5f : 0.0f;
                const float qx = ((float)(min_x + (int32_t)x) + off) / scale;
                const float qy = ((float)(max_y - (int32_t)y) - off) / scale;
                ((uint8_t *)out)[(size_t)y * out_stride + x] = sdf_value(pts, cstart, n_contours, qx, qy, scale);
            }
        return 0;
    }
    const int nn = n * n;
    for (uint32_t y = 0; y < h; ++y) {
        for (uint32_t x = 0; x < w; ++x) {
            int inside = 0;
            int16_t w00 = 0;
            for (int j = 0; j < n; ++j) {
                float py = (float)(max_y - (int32_t)y) - sub_offset(j, n, phase_center);
                float cy = py / scale;                                  /* render_glyph.zig:27 */
                for (int i = 0; i < n; ++i) {
                    float px = (float)(min_x + (int32_t)x) + sub_offset(i, n, phase_center);
                    float cx = px / scale;                              /* render_glyph.zig:26 */
                    int16_t wd = or_glyph_winding_at(pts, cstart, n_contours, cx, cy);
                    if (i == 0 && j == 0) w00 = wd;
                    inside += (wd != 0);
                }
            }
            size_t idx = (size_t)y * out_stride + x;
            switch (mode) {
            case OR_WINDING_I16:  ((int16_t *)out)[idx] = w00; break;
            case OR_GRAY_DEBUG:   ((uint8_t *)out)[idx] = gray_debug_value(w00); break;
            case OR_MASK_NONZERO: ((uint8_t *)out)[idx] = w00 != 0 ? 255 : 0; break; /* :29 */
            default:              ((uint8_t *)out)[idx] = (uint8_t)((2 * 255 * inside + nn) / (2 * nn)); break;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* BUILD-DEFINED signed distance field (no reference semantics: SURVEY F5).    */
/* CPU twin of font-renderer_amd/csrc/fr_sdf.hip — same operations, same order */
/* (written from the definition in that file's header comment, not shared).    */
/* ------------------------------------------------------------------------- */
static float sdf_seg_dist2(float p0x, float p0y, float p1x, float p1y, float p2x, float p2y, float qx, float qy)
{
    const float ax = p1x - p0x, ay = p1y - p0y;
    const float bx = p0x - 2.0f * p1x + p2x, by = p0y - 2.0f * p1y + p2y;
    float best = 3.402823466e+38f, bt = 0.0f;
    for (int k = 0; k <= 8; ++k) {                      /* 9 uniform probes */
        const float t = (float)k * 0.125f;
        const float x = p0x + 2.0f * t * ax + t * t * bx - qx;
        const float y = p0y + 2.0f * t * ay + t * t * by - qy;
        const float d2 = x * x + y * y;
        if (d2 < best) { best = d2; bt = t; }
    }
    const float lo = fmaxf(bt - 0.125f, 0.0f), hi = fminf(bt + 0.125f, 1.0f);
    float t = bt;
    for (int it = 0; it < 4; ++it) {                    /* 4 clamped Newton steps on (B-q).B' = 0 */
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

static uint8_t sdf_value(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours, float qx, float qy, float scale)
{
    float best = 3.402823466e+38f;
    int any = 0;
    for (uint32_t c = 0; c < n_contours; ++c) {
        const int16_t *cp = pts + 2 * (size_t)cstart[c];
        uint32_t curves = (cstart[c + 1] - cstart[c]) / 2;
        for (uint32_t k = 0; k < curves; ++k) {
            const int16_t *p = cp + 2 * (2 * k);
            float d2 = sdf_seg_dist2((float)p[0], (float)p[1], (float)p[2], (float)p[3], (float)p[4], (float)p[5], qx, qy);
            if (d2 < best) best = d2;
            any = 1;
        }
    }
    const int16_t w = or_glyph_winding_at(pts, cstart, n_contours, qx, qy);   /* sign: the reference's winding */
    float d = any ? sqrtf(best) * scale : 3.402823466e+38f;
    if (w == 0) d = -d;
    float v = 16.0f * d + 128.0f;
    v = floorf(v + 0.5f);
    v = fminf(fmaxf(v, 0.0f), 255.0f);
    return (uint8_t)v;
}

typedef struct {
    const int16_t *pts; const uint32_t *cstart; const uint32_t *gstart;
    const or_job *jobs; uint32_t n_jobs; int mode, n, phase_center;
    void *out; size_t out_stride; int tid, n_threads; int rc;
} batch_arg;

static void *batch_worker(void *p)
{
    batch_arg *a = (batch_arg *)p;
    size_t esz = a->mode == OR_WINDING_I16 ? 2 : 1;
    for (uint32_t j = (uint32_t)a->tid; j < a->n_jobs; j += (uint32_t)a->n_threads) {
        const or_job *jb = &a->jobs[j];
        uint32_t c0 = a->gstart[jb->glyph], c1 = a->gstart[jb->glyph + 1];
        char *dst = (char *)a->out + ((size_t)jb->out_y * a->out_stride + jb->out_x) * esz;
        int rc = or_render_cell(a->pts, a->cstart + c0, c1 - c0, jb->min_x, jb->max_y, jb->w, jb->h,
                                jb->scale, a->mode, a->n, a->phase_center, dst, a->out_stride);
        if (rc) a->rc = rc;
    }
    return NULL;
}

int or_render_batch(const int16_t *pts, const uint32_t *cstart, const uint32_t *gstart,
                    const or_job *jobs, uint32_t n_jobs, int mode, int n, int phase_center,
                    void *out, size_t out_stride, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    batch_arg args[256];
    for (int t = 0; t < n_threads; ++t) {
        batch_arg a = { pts, cstart, gstart, jobs, n_jobs, mode, n, phase_center, out, out_stride, t, n_threads, 0 };
        args[t] = a;
    }
    if (n_threads == 1) { batch_worker(&args[0]); return args[0].rc; }
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, batch_worker, &args[t]);
    int rc = 0;
    for (int t = 0; t < n_threads; ++t) { pthread_join(th[t], NULL); if (args[t].rc) rc = args[t].rc; }
    return rc;
}

/* ------------------------------------------------------------------------- */
/* a4: CurveType.init / GlyphInfo.init  (render_glyph.zig:96-146)              */
/* ------------------------------------------------------------------------- */
static int iabs(int v) { return v < 0 ? -v : v; }
static int isign(int v) { return (v > 0) - (v < 0); }

static uint8_t curve_type_i(int p0x, int p0y, int p1x, int p1y, int p2x, int p2y)
{
    if (p0y == p2y) return (p1y == p0y) ? OR_X_AXIS : OR_BALANCE;                       /* :97 */
    if (p0y < p2y) {                                                                    /* :98 */
        if (iabs(p0x + p2x - 2 * p1x) <= 1 && iabs(p0y + p2y - 2 * p1y) <= 1) return OR_UP_STRIGHT; /* :99 */
        if (p0y <= p1y && p1y <= p2y) return OR_UP_NORMAL;                              /* :100 */
        return (p1y < p0y) ? OR_UP_U : OR_UP_INV_U;                                     /* :101 */
    } else {                                                                            /* :102 */
        if (iabs(p0x + p2x - 2 * p1x) <= 1 && iabs(p0y + p2y - 2 * p1y) <= 1) return OR_DOWN_STRIGHT; /* :103 */
        if (p2y <= p1y && p1y <= p0y) return OR_DOWN_NORMAL;                            /* :104 */
        return (p1y > p0y) ? OR_DOWN_INV_U : OR_DOWN_U;                                 /* :105 */
    }
}

uint8_t or_curve_type(const int16_t p0[2], const int16_t p1[2], const int16_t p2[2])
{
    return curve_type_i(p0[0], p0[1], p1[0], p1[1], p2[0], p2[1]);
}

/* K = 1 is GlyphInfo.init; K > 1 applies the same rules to the glyph scaled by K (SURVEY §8 f-3,
 * build-defined: the reference has no such mode) */
static void glyph_info_init_k(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours, int K,
                              uint8_t *curve_type, uint8_t *include_p0)
{
    size_t o = 0;
    for (uint32_t c = 0; c < n_contours; ++c) {                                         /* :123 */
        const int16_t *cp = pts + 2 * (size_t)cstart[c];
        uint32_t len = cstart[c + 1] - cstart[c];
        uint32_t curve_count = len / 2;                                                 /* :124 */
        for (uint32_t k = 0; k < curve_count; ++k) {                                    /* :125 */
            const int16_t *p_2 = cp + 2 * (k != 0 ? 2 * k - 2 : len - 3);               /* :126 */
            const int16_t *p_1 = cp + 2 * (k != 0 ? 2 * k - 1 : len - 2);               /* :127 */
            const int16_t *p0 = cp + 2 * (2 * k);
            const int16_t *p1 = cp + 2 * (2 * k + 1);
            const int16_t *p2 = cp + 2 * (2 * k + 2);
            int prev_end = 2 * isign(K * p0[1] - K * p_1[1]) + isign(K * p0[1] - K * p_2[1]);   /* :132 */
            int curr_start = 2 * isign(K * p1[1] - K * p0[1]) + isign(K * p2[1] - K * p0[1]);   /* :133 */
            include_p0[o] = (uint8_t)(curr_start != 0 &&
                (prev_end == 0 || ((prev_end > 0) ^ (curr_start < 0))));                /* :136 */
            curve_type[o] = curve_type_i(K * p0[0], K * p0[1], K * p1[0], K * p1[1], K * p2[0], K * p2[1]); /* :137 */
            ++o;
        }
    }
}

void or_glyph_info_init(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                        uint8_t *curve_type, uint8_t *include_p0)
{
    glyph_info_init_k(pts, cstart, n_contours, 1, curve_type, include_p0);
}

/* ------------------------------------------------------------------------- */
/* a6/a7: exact predicates  (render_glyph.zig:249-300)                         */
/* The reference computes in i64; products such as dy*abxy*abxy can exceed i64  */
/* for long curves (safe builds trap, fast builds are UB).  The restatement     */
/* computes in __int128 — identical wherever the reference is defined — and     */
/* counts the out-of-domain events so tests can assert a fixture stays inside.  */
/* ------------------------------------------------------------------------- */
typedef __int128 i128;
static uint64_t g_ovf = 0, g_assert = 0;
uint64_t or_diag_i64_overflow_count(void) { return g_ovf; }
uint64_t or_diag_assert_fail_count(void) { return g_assert; }
void or_diag_reset(void) { g_ovf = 0; g_assert = 0; }

static i128 chk(i128 v)
{
    if (v > (i128)INT64_MAX || v < (i128)INT64_MIN) __atomic_fetch_add(&g_ovf, 1, __ATOMIC_RELAXED);
    return v;
}

int or_solve2_roots_winding(const int32_t p[2], const int32_t p0[2], const int32_t p1[2], const int32_t p2[2])
{
    i128 ay = (i128)p0[1] + p2[1] - 2 * (i128)p1[1];                /* :250 */
    i128 by = ((i128)p1[1] - p0[1]) * 2;                            /* :251 */
    i128 cy = (i128)p0[1] - p[1];                                   /* :252 */
    i128 dy = chk(by * by - 4 * ay * cy);                           /* :253 */
    if (dy <= 0) return 0;                                          /* :254 */
    i128 ax = (i128)p0[0] + p2[0] - 2 * (i128)p1[0];                /* :255 */
    i128 bx = ((i128)p1[0] - p0[0]) * 2;                            /* :256 */
    i128 cx = (i128)p0[0] - p[0];                                   /* :257 */
    i128 abxy = chk(ax * by - ay * bx);                             /* :267 */
    if (abxy == 0) return 0;                                        /* :268 */
    i128 tmp = chk(2 * ay * (ax * cy - ay * cx) - by * abxy);       /* :269 */
    if (tmp == 0) {                                                 /* :270 */
        return abxy > 0 ? 1 : -1;                                   /* :271 */
    } else if (tmp > 0) {                                           /* :272 */
        i128 l = chk(chk(dy * abxy) * abxy), r = chk(tmp * tmp);
        return (l < r) ? 0 : (abxy > 0 ? 1 : -1);                   /* :273 */
    } else {                                                        /* :274 */
        i128 l = chk(chk(dy * abxy) * abxy), r = chk(tmp * tmp);
        return (l <= r) ? 0 : (abxy > 0 ? 1 : -1);                  /* :275 */
    }
}

int or_solve1_root_crossing(const int32_t p[2], const int32_t p0[2], const int32_t p1[2],
                            const int32_t p2[2], int tilt_up)
{
    i128 ay = (i128)p0[1] + p2[1] - 2 * (i128)p1[1];                /* :280 */
    i128 by = ((i128)p1[1] - p0[1]) * 2;                            /* :281 */
    i128 cy = (i128)p0[1] - p[1];                                   /* :282 */
    i128 dy = chk(by * by - 4 * ay * cy);                           /* :283 */
    if (dy < 0) __atomic_fetch_add(&g_assert, 1, __ATOMIC_RELAXED); /* :284 assert(dy >= 0) */
    if (dy == 0) return 0;                                          /* :285 */
    i128 ax = (i128)p0[0] + p2[0] - 2 * (i128)p1[0];                /* :286 */
    i128 bx = ((i128)p1[0] - p0[0]) * 2;                            /* :287 */
    i128 cx = (i128)p0[0] - p[0];                                   /* :288 */
    i128 abxy = chk(ax * by - ay * bx);                             /* :290 */
    i128 tmp = chk(2 * ay * (ax * cy - ay * cx) - by * abxy);       /* :291 */
    if (abxy == 0) return tmp <= 0;                                 /* :292 */
    if ((abxy > 0) ^ (tilt_up != 0)) {                              /* :293 */
        if (tmp <= 0) return 1;                                     /* :294 */
        return chk(chk(dy * abxy) * abxy) >= chk(tmp * tmp);        /* :295 */
    } else {
        if (tmp >= 0) return 0;                                     /* :297 */
        return chk(chk(dy * abxy) * abxy) <= chk(tmp * tmp);        /* :298 */
    }
}

/* ------------------------------------------------------------------------- */
/* a5: windingInGlyph  (render_glyph.zig:160-247)                              */
/* ------------------------------------------------------------------------- */
static int16_t winding_in_glyph_k(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                                  const uint8_t *curve_type, const uint8_t *include_p0,
                                  int K, int32_t px, int32_t py)
{
    const int32_t p[2] = { px, py };                                           /* :161 */
    int16_t winding = 0;                                                       /* :163 */
    size_t o = 0;
    for (uint32_t c = 0; c < n_contours; ++c) {                                /* :164 */
        const int16_t *cp = pts + 2 * (size_t)cstart[c];
        uint32_t curve_count = (cstart[c + 1] - cstart[c]) / 2;
        for (uint32_t k = 0; k < curve_count; ++k, ++o) {                      /* :165 */
            const int32_t p0[2] = { K * cp[2 * (2 * k)], K * cp[2 * (2 * k) + 1] };           /* :166 */
            const int32_t p1[2] = { K * cp[2 * (2 * k + 1)], K * cp[2 * (2 * k + 1) + 1] };   /* :167 */
            const int32_t p2[2] = { K * cp[2 * (2 * k + 2)], K * cp[2 * (2 * k + 2) + 1] };   /* :168 */
            const int inc = include_p0[o] != 0;
            switch (curve_type[o]) {                                           /* :170 */
            case OR_X_AXIS: break;                                             /* :171 */
            case OR_BALANCE:                                                   /* :172 */
                if (inc && p[1] == p0[1]) {                                    /* :173 */
                    if (p[0] < p0[0]) winding += (p1[1] < p0[1]) ? 1 : -1;     /* :174 */
                } else if ((!(p1[1] < p0[1])) ^ (p[1] < p0[1])) {              /* :175 */
                    winding += (int16_t)or_solve2_roots_winding(p, p0, p1, p2);/* :176 */
                }
                break;
            case OR_UP_STRIGHT:                                                /* :179 */
                if ((p0[1] < p[1] || (inc && p0[1] == p[1])) && p[1] < p2[1]) {/* :180 */
                    int64_t v1 = (int64_t)(p[1] - p0[1]) * (p2[0] - p0[0]);    /* :181 */
                    int64_t v2 = (int64_t)(p[0] - p0[0]) * (p2[1] - p0[1]);    /* :182 */
                    if (v1 >= v2) winding += -1;                               /* :183 */
                }
                break;
            case OR_UP_NORMAL:                                                 /* :186 */
                if ((p0[1] < p[1] || (inc && p0[1] == p[1])) && p[1] < p2[1]) {/* :187 */
                    if (or_solve1_root_crossing(p, p0, p1, p2, 1)) winding += -1; /* :188 */
                }
                break;
            case OR_UP_U:                                                      /* :191 */
                if (p0[1] <= p[1] && p[1] < p2[1]) {                           /* :192 */
                    int cross = or_solve1_root_crossing(p, p0, p1, p2, 1);     /* :193 */
                    if (p0[1] < p[1]) {                                        /* :194 */
                        if (cross) winding += -1;                              /* :195 */
                    } else if (inc && (cross ^ (p[0] <= p0[0]))) {             /* :196 */
                        winding += cross ? -1 : 1;                             /* :197 */
                    }
                } else if (p[1] < p0[1]) {                                     /* :199 */
                    winding += (int16_t)or_solve2_roots_winding(p, p0, p1, p2);/* :200 */
                }
                break;
            case OR_UP_INV_U:                                                  /* :203 */
                if ((p0[1] < p[1] || (inc && p0[1] == p[1])) && p[1] <= p2[1]) {/* :204 */
                    if (or_solve1_root_crossing(p, p0, p1, p2, 1)) winding += -1; /* :205 */
                } else if (p2[1] < p[1]) {                                     /* :206 */
                    winding += (int16_t)or_solve2_roots_winding(p, p0, p1, p2);/* :207 */
                }
                break;
            case OR_DOWN_STRIGHT:                                              /* :210 */
                if (p2[1] < p[1] && (p[1] < p0[1] || (inc && p[1] == p0[1]))) {/* :211 */
                    int64_t v1 = (int64_t)(p[1] - p0[1]) * (p2[0] - p0[0]);    /* :212 */
                    int64_t v2 = (int64_t)(p[0] - p0[0]) * (p2[1] - p0[1]);    /* :213 */
                    if (v1 <= v2) winding += 1;                                /* :214 */
                }
                break;
            case OR_DOWN_NORMAL:                                               /* :217 */
                if (p2[1] < p[1] && (p[1] < p0[1] || (inc && p[1] == p0[1]))) {/* :218 */
                    if (or_solve1_root_crossing(p, p0, p1, p2, 0)) winding += 1; /* :219 */
                }
                break;
            case OR_DOWN_INV_U:                                                /* :222 */
                if (p2[1] < p[1] && p[1] <= p0[1]) {                           /* :223 */
                    int cross = or_solve1_root_crossing(p, p0, p1, p2, 0);     /* :224 */
                    if (p0[1] > p[1]) {                                        /* :225 */
                        if (cross) winding += -1;                              /* :226 (sic: -1) */
                    } else if (inc && (cross ^ (p[0] > p0[0]))) {              /* :227 */
                        winding += cross ? 1 : -1;                             /* :228 */
                    }
                } else if (p[1] > p0[1]) {                                     /* :230 */
                    winding += (int16_t)or_solve2_roots_winding(p, p0, p1, p2);/* :231 */
                }
                break;
            case OR_DOWN_U:                                                    /* :234 */
                if (p2[1] <= p[1] && (p[1] < p0[1] || (inc && p[1] == p0[1]))) {/* :235 */
                    if (or_solve1_root_crossing(p, p0, p1, p2, 0)) winding += 1; /* :236-237 */
                } else if (p[1] < p2[1]) {                                     /* :239 */
                    winding += (int16_t)or_solve2_roots_winding(p, p0, p1, p2);/* :240 */
                }
                break;
            default: break;
            }
        }
    }
    return winding;                                                            /* :246 */
}

int16_t or_winding_in_glyph(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                            const uint8_t *curve_type, const uint8_t *include_p0,
                            int16_t px, int16_t py)
{
    return winding_in_glyph_k(pts, cstart, n_contours, curve_type, include_p0, 1, px, py);
}

static size_t total_curves(const uint32_t *cstart, uint32_t n_contours)
{
    size_t n = 0;
    for (uint32_t c = 0; c < n_contours; ++c) n += (cstart[c + 1] - cstart[c]) / 2;
    return n;
}

/* Image.zig:227-236: lattice of GlyphDebug.render */
void or_winding_lattice(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                        const int16_t box[4], int16_t *out)
{
    size_t nc = total_curves(cstart, n_contours);
    uint8_t *ct = (uint8_t *)malloc(nc ? nc : 1), *ip = (uint8_t *)malloc(nc ? nc : 1);
    or_glyph_info_init(pts, cstart, n_contours, ct, ip);                       /* Image.zig:221 */
    uint32_t W = (uint32_t)(box[2] - box[0] + 3), H = (uint32_t)(box[3] - box[1] + 3); /* Image.zig:183 */
    for (uint32_t h = 0; h < H; ++h)
        for (uint32_t w = 0; w < W; ++w) {
            int16_t y = (int16_t)(box[3] - (int16_t)h + 1);                    /* Image.zig:232 */
            int16_t x = (int16_t)(box[0] + (int16_t)w - 1);                    /* Image.zig:233 */
            out[(size_t)h * W + w] = or_winding_in_glyph(pts, cstart, n_contours, ct, ip, x, y);
        }
    free(ct); free(ip);
}

/* SURVEY §8 f-3 (BUILD-DEFINED twin of fr_exact_lattice / fr_exact_coverage): a4-a7 applied to the
 * glyph scaled by K, at the integer points (x0 + i, y0 - j) of the scaled glyph */
void or_exact_lattice(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours, uint32_t K,
                      int32_t x0, int32_t y0, uint32_t w, uint32_t h, int16_t *out)
{
    size_t nc = total_curves(cstart, n_contours);
    uint8_t *ct = (uint8_t *)malloc(nc ? nc : 1), *ip = (uint8_t *)malloc(nc ? nc : 1);
    glyph_info_init_k(pts, cstart, n_contours, (int)K, ct, ip);
    for (uint32_t j = 0; j < h; ++j)
        for (uint32_t i = 0; i < w; ++i)
            out[(size_t)j * w + i] = winding_in_glyph_k(pts, cstart, n_contours, ct, ip, (int)K,
                                                        x0 + (int32_t)i, y0 - (int32_t)j);
    free(ct); free(ip);
}

void or_exact_coverage(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours, uint32_t K,
                       int32_t x0, int32_t y0, uint32_t w_px, uint32_t h_px, uint32_t n, uint8_t *out)
{
    size_t W = (size_t)w_px * n, H = (size_t)h_px * n;
    int16_t *lat = (int16_t *)malloc((W * H ? W * H : 1) * sizeof(int16_t));
    or_exact_lattice(pts, cstart, n_contours, K, x0, y0, (uint32_t)W, (uint32_t)H, lat);
    for (uint32_t y = 0; y < h_px; ++y)
        for (uint32_t x = 0; x < w_px; ++x) {
            uint32_t inside = 0;
            for (uint32_t j = 0; j < n; ++j)
                for (uint32_t i = 0; i < n; ++i) inside += lat[((size_t)y * n + j) * W + (size_t)x * n + i] != 0;
            out[(size_t)y * w_px + x] = (uint8_t)((2u * 255u * inside + n * n) / (2u * n * n));
        }
    free(lat);
}

/* ------------------------------------------------------------------------- */
/* a8: Image sinks                                                             */
/* ------------------------------------------------------------------------- */
void or_gray_rgb(uint8_t val, uint8_t rgb[3]) { rgb[0] = rgb[1] = rgb[2] = val; }   /* Image.zig:78-82 */

static void winding_color(int16_t val, uint8_t scaler, uint8_t overflow_color, uint8_t rgb[3])
{
    /* Image.zig:121-129 and :192-200 share this arithmetic: u16 saturating multiply */
    uint32_t a = (uint32_t)(val < 0 ? -(int32_t)val : val);      /* @abs(i16) -> u16 */
    uint32_t c = a * (uint32_t)scaler;
    if (c > 65535u) c = 65535u;                                   /* *| saturates at u16 */
    uint8_t color = (uint8_t)(c > 255u ? 255u : c);               /* truncate(clamp(c,0,255)) */
    uint8_t sub = (c == color) ? 0 : overflow_color;
    if (val > 0) { rgb[0] = sub; rgb[1] = sub; rgb[2] = color; }
    else { rgb[0] = color; rgb[1] = sub; rgb[2] = sub; }
}

void or_winding_rgb(int16_t val, uint8_t scaler, uint8_t overflow_color, uint8_t rgb[3])
{
    if (val == 0) { rgb[0] = rgb[1] = rgb[2] = 0; return; }       /* Image.zig:124 */
    winding_color(val, scaler, overflow_color, rgb);
}

/* Image.zig:220-240 GlyphDebug.render (init :181-190 with overflow 150, on {255,255,0}, off {0,255,255}) */
void or_glyph_debug_render(const int16_t *pts, const uint32_t *cstart, uint32_t n_contours,
                           const int16_t box[4], uint8_t winding_scale, uint8_t *rgb_out)
{
    uint32_t W = (uint32_t)(box[2] - box[0] + 3), H = (uint32_t)(box[3] - box[1] + 3);
    int16_t *lat = (int16_t *)malloc(sizeof(int16_t) * (size_t)W * H);
    or_winding_lattice(pts, cstart, n_contours, box, lat);
    for (size_t i = 0; i < (size_t)W * H; ++i)
        winding_color(lat[i], winding_scale, 150, &rgb_out[3 * i]);   /* setWindingLinear :192-200
            (the early "== 0" store at :193-195 is overwritten by :199 with the same {0,0,0}) */
    free(lat);
    for (uint32_t c = 0; c < n_contours; ++c) {                        /* setGlyphPoints :202-218 */
        const int16_t *cp = pts + 2 * (size_t)cstart[c];
        uint32_t curve_count = (cstart[c + 1] - cstart[c]) / 2;
        for (uint32_t k = 0; k < curve_count; ++k) {
            const int16_t *on = cp + 2 * (2 * k), *off = cp + 2 * (2 * k + 1);
            uint32_t w0 = (uint32_t)(on[0] - box[0] + 1), h0 = (uint32_t)(box[3] - on[1] + 1);
            uint8_t *d = &rgb_out[3 * ((size_t)h0 * W + w0)];
            d[0] = 255; d[1] = 255; d[2] = 0;
            uint32_t w1 = (uint32_t)(off[0] - box[0] + 1), h1 = (uint32_t)(box[3] - off[1] + 1);
            d = &rgb_out[3 * ((size_t)h1 * W + w1)];
            d[0] = 0; d[1] = 255; d[2] = 255;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* f2: QOI writer  (src/tools/qoi.zig:25-88)                                   */
/* ------------------------------------------------------------------------- */
static uint8_t running_index(uint8_t r, uint8_t g, uint8_t b, uint8_t a)       /* qoi.zig:85-88 */
{
    return (uint8_t)((uint8_t)(r * 3) + (uint8_t)(g * 5) + (uint8_t)(b * 7) + (uint8_t)(a * 11)) & 63;
}

size_t or_qoi_encode_rgb(const uint8_t *rgb, uint32_t w, uint32_t h, uint8_t *out, size_t cap)
{
    size_t total = (size_t)w * h, o = 0;
#define PUT(b) do { if (o >= cap) return 0; out[o++] = (uint8_t)(b); } while (0)
    PUT('q'); PUT('o'); PUT('i'); PUT('f');                                     /* :26-32 header, big-endian */
    PUT(w >> 24); PUT(w >> 16); PUT(w >> 8); PUT(w);
    PUT(h >> 24); PUT(h >> 16); PUT(h >> 8); PUT(h);
    PUT(3); PUT(0);
    uint8_t running[64][3];
    memset(running, 0, sizeof running);                                         /* :35 */
    uint8_t prev[3] = { 0, 0, 0 }, curr[3] = { 0, 0, 0 };                       /* :37-38 */
    uint8_t ridx = 0;                                                           /* :39 */
    size_t next_idx = 0;
    while (next_idx < total) {                                                  /* :41 */
        memcpy(running[ridx], curr, 3);                                         /* :42 */
        memcpy(prev, curr, 3);                                                  /* :43 */
        memcpy(curr, &rgb[3 * next_idx], 3);                                    /* :44 */
        ridx = running_index(curr[0], curr[1], curr[2], 255);                   /* :45 */
        next_idx += 1;
        if (memcmp(prev, curr, 3) == 0) {                                       /* :48 */
            uint8_t run = 0;
            while (run < 0x3D && next_idx < total && memcmp(prev, &rgb[3 * next_idx], 3) == 0) { /* :50-54 */
                next_idx += 1; run += 1;
            }
            PUT(0xC0 | run);                                                    /* :58 */
            continue;
        } else if (memcmp(running[ridx], curr, 3) == 0) {                       /* :61 */
            PUT(0x00 | ridx);                                                   /* :62 */
            continue;
        } else {
            uint8_t dr = (uint8_t)(curr[0] - prev[0] + 2);                      /* :67 */
            uint8_t dg = (uint8_t)(curr[1] - prev[1] + 2);
            uint8_t db = (uint8_t)(curr[2] - prev[2] + 2);
            if (dr < 4 && dg < 4 && db < 4) {                                   /* :68 */
                PUT(0x40 | (dr << 4) | (dg << 2) | db);                         /* :69 */
                continue;
            }
            dr = (uint8_t)(dr + (uint8_t)(8 - dg));                             /* :73 */
            db = (uint8_t)(db + (uint8_t)(8 - dg));
            dg = (uint8_t)(dg + 30);
            if (dr < 16 && dg < 64 && db < 16) {                                /* :74 */
                PUT(0x80 | dg); PUT((dr << 4) | db);                            /* :75 */
                continue;
            }
        }
        PUT(0xFE); PUT(curr[0]); PUT(curr[1]); PUT(curr[2]);                    /* :80 */
    }
    PUT(0); PUT(0); PUT(0); PUT(0); PUT(0); PUT(0); PUT(0); PUT(1);             /* :82 */
#undef PUT
    return o;
}
