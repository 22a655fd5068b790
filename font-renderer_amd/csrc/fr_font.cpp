// fr_font.cpp — the contour producer on the C side: TrueType glyf/loca walk ->
// the Glyph contour layout the rasterizer consumes, so whole fonts can be batch-fed without the
// Zig host.  Host code (the reference does this on the CPU too); no GPU involved.
//
// Follows, statement by statement where behaviour is observable:
//   /root/reference/src/font/Font.zig:31-85    initTTF (table directory, head, maxp, cmap, loca, glyf)
//   /root/reference/src/font/Font.zig:171-221  loadGlyph (loca short/long, simple / composite,
//                                              dependency-loop check, cache)
//   /root/reference/src/font/ttf.zig:759-817   SimpleGlyph.initFromReader (flags, repeat, x/y deltas)
//   /root/reference/src/font/ttf.zig:830-929   ComponentGlyph (+ PartDescription) parsing
//   /root/reference/src/font/Glyph.zig:26-106  countTTFPoints / Contour.initTTF / initTTFSimple
//   /root/reference/src/font/Glyph.zig:108-182 initTTFComponent / transform1 (2.14 fixed point)
//   /root/reference/src/tools/geometry.zig:12-17 Point.initMiddle (truncating midpoint)
//   /root/reference/src/font/ttf.zig:433-449, 608-614  cmap format 4 / 12 glyphIndex
// Where the reference @panic("not impl")s (hinting instructions Glyph.zig:85,109; use_my_metrics
// :110; point-matching args :134; fractional offsets without round_xy_to_grid :146-147) this
// returns FR_E_UNSUPPORTED — except that instructions may be tolerated on request
// (FR_FONT_ALLOW_HINTED: the outline itself does not depend on them).
#include "../../include/fr_raster.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

extern "C" const char *fr_last_error(void);
namespace fr { int set_error(int code, const char *fmt, ...); }

namespace {

struct GlyphData {
    bool loaded = false;
    int16_t box[4] = {0, 0, 0, 0};
    std::vector<int16_t> pts;              // (x, y) pairs, contours back to back
    std::vector<uint32_t> cstart{0};       // contour offsets in points
};

}  // namespace

struct fr_font {
    std::vector<uint8_t> d;
    uint16_t units_per_em = 0, num_glyphs = 0;
    bool y0_baseline = false, loca_long = false, allow_hinted = false;
    size_t pos_loca = 0, pos_glyf = 0, pos_cmap_sub = 0, pos_hmtx = 0;
    uint16_t num_long_hor_metrics = 0;
    int cmap_format = 0;
    std::vector<GlyphData> glyphs;
    size_t composite_depth = 0;     // load_composite frames currently on the stack

    bool has(size_t off, size_t n) const { return off <= d.size() && n <= d.size() - off; }
    uint16_t u16(size_t o) const { return (uint16_t)((d[o] << 8) | d[o + 1]); }
    int16_t i16(size_t o) const { return (int16_t)u16(o); }
    uint32_t u32(size_t o) const { return ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]; }
};

namespace {

using fr::set_error;

constexpr uint32_t tag(const char (&s)[5]) { return ((uint32_t)s[0] << 24) | ((uint32_t)s[1] << 16) | ((uint32_t)s[2] << 8) | (uint32_t)s[3]; }

// ---- cmap encoding-subtable ranking, ttf.zig:258-312
struct Enc { uint16_t platform, specific; uint32_t offset; };
bool is_unicode(const Enc &e) { return e.platform == 0 ? e.specific != 14 : (e.platform == 3 ? (e.specific == 1 || e.specific == 10) : false); }
int bmp_restriction(const Enc &e)   // 0 unknown, 1 true, 2 false
{
    if (e.platform == 0) return e.specific == 3 ? 1 : ((e.specific == 4 || e.specific == 6) ? 2 : 0);
    if (e.platform == 3) return e.specific == 1 ? 1 : (e.specific == 10 ? 2 : 0);
    return 0;
}
bool is_discarded(const Enc &e) { return e.platform == 0 && e.specific == 2; }
bool is_the_best(const Enc &e) { return is_unicode(e) && !is_discarded(e) && bmp_restriction(e) == 2; }
bool is_better_than(const Enc &a, const Enc &b)
{
    if (!is_unicode(a)) return false;
    if (!is_unicode(b)) return true;
    if (is_discarded(a)) return false;
    if (is_discarded(b)) return true;
    return bmp_restriction(a) >= bmp_restriction(b);
}

// geometry.zig:12-17: @divTrunc(a + b, 2)
inline int16_t mid(int16_t a, int16_t b) { return (int16_t)(((int)a + (int)b) / 2); }

// Glyph.zig:178-182
int32_t transform1(int16_t x, int16_t y, int16_t a_or_b, int16_t c_or_d, int16_t e_or_f)
{
    const int32_t aa = a_or_b < 0 ? -(int32_t)a_or_b : a_or_b, cc = c_or_d < 0 ? -(int32_t)c_or_d : c_or_d;
    const int32_t tmp = aa > cc ? aa : cc;                                         // :179
    const int16_t diff = (int16_t)(uint16_t)((uint16_t)aa - (uint16_t)cc);
    const int32_t dabs = diff < 0 ? -(int32_t)diff : diff;
    const int16_t shift = (int16_t)((int32_t)e_or_f * (dabs <= 8 ? 2 : 1));         // :180 (a shift of a negative offset, written as a product)
    return (int32_t)a_or_b * x + (int32_t)c_or_d * y + tmp * (int32_t)shift;       // :181
}
int32_t round14(int32_t d)          // helpers.zig:112-123, bias_bits = 14
{
    const int32_t base = d >> 14;
    if (d < 0) {
        if (d == INT32_MIN) return base;
        return ((-d) & (1 << 13)) == 0 ? base : base - 1;
    }
    return (d & (1 << 13)) == 0 ? base : base + 1;
}

int load_glyph(fr_font &f, uint16_t gi, std::vector<uint16_t> &track);
constexpr size_t kMaxCompositeDepth = 64;          // nesting of composite glyphs (the reference recurses unbounded: Font.zig:204-216)
constexpr size_t kMaxGlyphPoints = 1u << 20;       // expanded points of one glyph

int load_simple(fr_font &f, GlyphData &g, size_t pos, uint16_t n_contours)
{
    // ttf.zig:759-817
    if (!f.has(pos, 2u * n_contours + 2u)) return set_error(FR_E_INVALID, "glyf: truncated endPtsOfContours");
    std::vector<uint16_t> ends(n_contours);
    for (uint16_t c = 0; c < n_contours; ++c) {
        ends[c] = f.u16(pos + 2u * c);
        if (c && ends[c] <= ends[c - 1]) return set_error(FR_E_INVALID, "glyf: endPtsOfContours not increasing");   // ensureMonoIncrease :766
    }
    pos += 2u * n_contours;
    const uint32_t point_count = (uint32_t)ends[n_contours - 1] + 1;                 // :767
    const uint16_t ins_len = f.u16(pos);                                            // :769
    pos += 2;
    if (ins_len > 0 && !f.allow_hinted)
        return set_error(FR_E_UNSUPPORTED, "glyph carries %u bytes of instructions (the reference panics: Glyph.zig:85)", ins_len);
    if (!f.has(pos, ins_len)) return set_error(FR_E_INVALID, "glyf: truncated instructions");
    pos += ins_len;
    std::vector<uint8_t> flags;
    flags.reserve(point_count);
    while (flags.size() < point_count) {                                             // :775-783
        if (!f.has(pos, 1)) return set_error(FR_E_INVALID, "glyf: truncated flags");
        const uint8_t fl = f.d[pos++];
        if (fl & 8) {
            if (!f.has(pos, 1)) return set_error(FR_E_INVALID, "glyf: truncated flags");
            const uint16_t rep = (uint16_t)f.d[pos++] + 1;
            flags.insert(flags.end(), rep, fl);
        } else {
            flags.push_back(fl);
        }
    }
    if (flags.size() != point_count) return set_error(FR_E_INVALID, "glyf: flag repeat overruns the point count");   // assert :784
    std::vector<int16_t> xs(point_count), ys(point_count);
    int16_t acc = 0;                                                                 // :794-803
    for (uint32_t i = 0; i < point_count; ++i) {
        const uint8_t fl = flags[i];
        if (fl & 2) {
            if (!f.has(pos, 1)) return set_error(FR_E_INVALID, "glyf: truncated x coordinates");
            const uint8_t v = f.d[pos++];
            acc = (int16_t)(acc + ((fl & 16) ? (int16_t)v : (int16_t)-(int16_t)v));
        } else if (!(fl & 16)) {
            if (!f.has(pos, 2)) return set_error(FR_E_INVALID, "glyf: truncated x coordinates");
            acc = (int16_t)(acc + f.i16(pos));
            pos += 2;
        }
        xs[i] = acc;
    }
    acc = 0;                                                                         // :805-814
    for (uint32_t i = 0; i < point_count; ++i) {
        const uint8_t fl = flags[i];
        if (fl & 4) {
            if (!f.has(pos, 1)) return set_error(FR_E_INVALID, "glyf: truncated y coordinates");
            const uint8_t v = f.d[pos++];
            acc = (int16_t)(acc + ((fl & 32) ? (int16_t)v : (int16_t)-(int16_t)v));
        } else if (!(fl & 32)) {
            if (!f.has(pos, 2)) return set_error(FR_E_INVALID, "glyf: truncated y coordinates");
            acc = (int16_t)(acc + f.i16(pos));
            pos += 2;
        }
        ys[i] = acc;
    }
    // Glyph.zig:84-106 initTTFSimple -> Contour.initTTF per contour (:43-74)
    uint32_t start = 0;
    for (uint16_t c = 0; c < n_contours; ++c) {
        const uint32_t end = ends[c];
        const size_t base = g.pts.size() / 2;
        bool prev_on = (flags[end] & 1) != 0;                                        // :47
        int16_t px = xs[end], py = ys[end];                                          // :48
        if (prev_on) { g.pts.push_back(0); g.pts.push_back(0); }                     // next_point starts at 1 (:49)
        for (uint32_t i = start; i <= end; ++i) {                                    // :50
            const bool on = (flags[i] & 1) != 0;
            if (prev_on == on) {                                                     // :54-58 implied point
                g.pts.push_back(mid(px, xs[i]));
                g.pts.push_back(mid(py, ys[i]));
            }
            g.pts.push_back(xs[i]);                                                  // :60
            g.pts.push_back(ys[i]);
            prev_on = on; px = xs[i]; py = ys[i];
        }
        if (prev_on) {                                                               // :67-68
            g.pts[2 * base] = g.pts[g.pts.size() - 2];
            g.pts[2 * base + 1] = g.pts[g.pts.size() - 1];
        } else {                                                                     // :69-71
            const int16_t fx = g.pts[2 * base], fy = g.pts[2 * base + 1];
            g.pts.push_back(fx);
            g.pts.push_back(fy);
        }
        g.cstart.push_back((uint32_t)(g.pts.size() / 2));
        start = end + 1;
    }
    return FR_OK;
}

int load_composite(fr_font &f, GlyphData &g, size_t pos, uint16_t self_index, std::vector<uint16_t> &track)
{
    struct Part { uint16_t flag, glyph, a1, a2; int16_t m[4]; };
    std::vector<Part> parts;
    for (;;) {                                                                       // ttf.zig:894-903
        if (!f.has(pos, 4)) return set_error(FR_E_INVALID, "glyf: truncated component");
        Part p{};
        p.flag = f.u16(pos); p.glyph = f.u16(pos + 2);
        pos += 4;
        if (p.flag & 1) {                                                            // arg_1_and_arg_2_are_words :863
            if (!f.has(pos, 4)) return set_error(FR_E_INVALID, "glyf: truncated component");
            p.a1 = f.u16(pos); p.a2 = f.u16(pos + 2); pos += 4;
        } else {                                                                     // bytes, zero-extended as the reference does (:866)
            if (!f.has(pos, 2)) return set_error(FR_E_INVALID, "glyf: truncated component");
            p.a1 = f.d[pos]; p.a2 = f.d[pos + 1]; pos += 2;
        }
        const int16_t one = 1 << 14;
        if (p.flag & 0x0008) {                                                       // we_have_a_scale :869
            if (!f.has(pos, 2)) return set_error(FR_E_INVALID, "glyf: truncated component");
            const int16_t s = f.i16(pos); pos += 2;
            p.m[0] = s; p.m[1] = 0; p.m[2] = 0; p.m[3] = s;
        } else if (p.flag & 0x0040) {                                                // x and y scale :874
            if (!f.has(pos, 4)) return set_error(FR_E_INVALID, "glyf: truncated component");
            p.m[0] = f.i16(pos); p.m[1] = 0; p.m[2] = 0; p.m[3] = f.i16(pos + 2); pos += 4;
        } else if (p.flag & 0x0080) {                                                // two by two :878
            if (!f.has(pos, 8)) return set_error(FR_E_INVALID, "glyf: truncated component");
            for (int k = 0; k < 4; ++k) p.m[k] = f.i16(pos + 2 * k);
            pos += 8;
        } else {
            p.m[0] = one; p.m[1] = 0; p.m[2] = 0; p.m[3] = one;
        }
        parts.push_back(p);
        if (!(p.flag & 0x0020)) break;                                               // more_components :902
        if (parts.size() > 4096) return set_error(FR_E_INVALID, "glyf: runaway component list");
    }
    if (parts.back().flag & 0x0100) {                                                // we_have_instructions :907
        if (!f.has(pos, 2)) return set_error(FR_E_INVALID, "glyf: truncated component instructions");
        if (f.u16(pos) > 0 && !f.allow_hinted)
            return set_error(FR_E_UNSUPPORTED, "composite glyph carries instructions (the reference panics: Glyph.zig:109)");
    }
    for (const Part &p : parts)
        if (p.flag & 0x0200) return set_error(FR_E_UNSUPPORTED, "use_my_metrics component (the reference panics: Glyph.zig:110)");
    // Font.zig:204-216: dependency-loop check (the stack is never popped in the reference), load parts
    // (bounds the reference lacks: nesting depth and expanded size — a crafted chain of distinct
    // composites would otherwise recurse ~65k deep or grow exponentially)
    struct Depth { size_t &d; explicit Depth(size_t &x) : d(x) { ++d; } ~Depth() { --d; } } depth(f.composite_depth);
    if (f.composite_depth > kMaxCompositeDepth)
        return set_error(FR_E_UNSUPPORTED, "composite glyphs nested deeper than %u", (unsigned)kMaxCompositeDepth);
    track.push_back(self_index);
    for (const Part &p : parts) {
        for (uint16_t seen : track)
            if (seen == p.glyph) return set_error(FR_E_INVALID, "loop component glyph dependencies");
        if (p.glyph >= f.num_glyphs) return set_error(FR_E_INVALID, "component glyph index %u out of range", p.glyph);
        if (!f.glyphs[p.glyph].loaded) {
            int rc = load_glyph(f, p.glyph, track);
            if (rc) return rc;
        }
    }
    // Glyph.zig:108-165 initTTFComponent
    for (const Part &p : parts) {
        if (!(p.flag & 0x0002)) return set_error(FR_E_UNSUPPORTED, "component placed by point matching (the reference panics: Glyph.zig:134)");
        const GlyphData &src = f.glyphs[p.glyph];
        for (size_t c = 0; c + 1 < src.cstart.size(); ++c) {
            for (uint32_t i = src.cstart[c]; i < src.cstart[c + 1]; ++i) {
                const int16_t x = src.pts[2 * i], y = src.pts[2 * i + 1];
                const int32_t fx = transform1(x, y, p.m[0], p.m[2], (int16_t)p.a1);  // :138
                const int32_t fy = transform1(x, y, p.m[1], p.m[3], (int16_t)p.a2);  // :139
                int32_t ox, oy;
                if (p.flag & 0x0004) {                                               // round_xy_to_grid :141
                    ox = round14(fx); oy = round14(fy);
                } else {
                    ox = fx >> 14; oy = fy >> 14;                                    // :144-145
                    if (ox * 16384 != fx || oy * 16384 != fy)
                        return set_error(FR_E_UNSUPPORTED, "fractional component offset without round_xy_to_grid (the reference panics: Glyph.zig:146)");
                }
                g.pts.push_back((int16_t)ox);
                g.pts.push_back((int16_t)oy);
            }
            if (g.pts.size() / 2 > kMaxGlyphPoints)
                return set_error(FR_E_UNSUPPORTED, "composite glyph expands to more than %u points", (unsigned)kMaxGlyphPoints);
            g.cstart.push_back((uint32_t)(g.pts.size() / 2));
        }
    }
    return FR_OK;
}

int load_glyph(fr_font &f, uint16_t gi, std::vector<uint16_t> &track)
{
    GlyphData &g = f.glyphs[gi];
    g = GlyphData{};
    // Font.zig:173-192
    uint32_t off0, off1;
    if (f.loca_long) {
        if (!f.has(f.pos_loca + 4u * gi, 8)) return set_error(FR_E_INVALID, "loca: truncated");
        off0 = f.u32(f.pos_loca + 4u * gi); off1 = f.u32(f.pos_loca + 4u * gi + 4);
    } else {
        if (!f.has(f.pos_loca + 2u * gi, 4)) return set_error(FR_E_INVALID, "loca: truncated");
        off0 = 2u * f.u16(f.pos_loca + 2u * gi); off1 = 2u * f.u16(f.pos_loca + 2u * gi + 2);
    }
    if (off0 == off1) { g.loaded = true; return FR_OK; }                             // Glyph.initEmpty
    const size_t pos = f.pos_glyf + off0;
    if (!f.has(pos, 10)) return set_error(FR_E_INVALID, "glyf: truncated glyph description");
    const int16_t nc = f.i16(pos);                                                   // :195
    if (nc == 0) { g.loaded = true; return FR_OK; }                                  // :196-197
    for (int k = 0; k < 4; ++k) g.box[k] = f.i16(pos + 2 + 2 * k);
    int rc = nc > 0 ? load_simple(f, g, pos + 10, (uint16_t)nc) : load_composite(f, g, pos + 10, gi, track);
    if (rc) { g = GlyphData{}; return rc; }
    g.loaded = true;
    return FR_OK;
}

int ensure_loaded(fr_font *f, uint16_t gi)
{
    if (!f) return set_error(FR_E_INVALID, "font is NULL");
    if (gi >= f->num_glyphs) return set_error(FR_E_INVALID, "glyph index %u of %u", gi, f->num_glyphs);
    if (f->glyphs[gi].loaded) return FR_OK;
    std::vector<uint16_t> track;                                                     // Font.zig:164
    return load_glyph(*f, gi, track);
}

}  // namespace

extern "C" {

int fr_font_open(const void *data, size_t len, uint32_t flags, fr_font **out)
{
    if (!data || !out) return set_error(FR_E_INVALID, "fr_font_open: NULL argument");
    *out = nullptr;
    fr_font *f = new (std::nothrow) fr_font;
    if (!f) return set_error(FR_E_NOMEM, "fr_font_open: allocation");
    f->d.assign((const uint8_t *)data, (const uint8_t *)data + len);
    f->allow_hinted = (flags & FR_FONT_ALLOW_HINTED) != 0;
    auto bail = [&](int rc) { delete f; return rc; };
    if (!f->has(0, 12)) return bail(set_error(FR_E_INVALID, "not a TrueType file"));
    const uint16_t n_tables = f->u16(4);                                             // OffsetSubtable, Font.zig:38
    if (!f->has(12, 16u * n_tables)) return bail(set_error(FR_E_INVALID, "truncated table directory"));
    size_t head = 0, maxp = 0, cmap = 0, loca = 0, glyf = 0, hhea = 0, hmtx = 0;
    for (uint16_t t = 0; t < n_tables; ++t) {
        const size_t e = 12 + 16u * t;
        const uint32_t tg = f->u32(e), off = f->u32(e + 8);
        if (tg == tag("head")) head = off;
        else if (tg == tag("maxp")) maxp = off;
        else if (tg == tag("cmap")) cmap = off;
        else if (tg == tag("loca")) loca = off;
        else if (tg == tag("glyf")) glyf = off;
        else if (tg == tag("hhea")) hhea = off;
        else if (tg == tag("hmtx")) hmtx = off;
    }
    if (!head || !maxp || !cmap || !loca || !glyf) return bail(set_error(FR_E_INVALID, "missing head/maxp/cmap/loca/glyf table"));   // Font.zig:43-59
    if (!f->has(head, 54) || !f->has(maxp, 6)) return bail(set_error(FR_E_INVALID, "truncated head/maxp"));
    f->y0_baseline = (f->u16(head + 16) & 1) != 0;
    f->units_per_em = f->u16(head + 18);
    f->loca_long = f->i16(head + 50) != 0;
    f->num_glyphs = f->u16(maxp + 4);
    f->pos_loca = loca; f->pos_glyf = glyf;
    // hhea / hmtx: Font.zig:64-69 (both required there too)
    if (!hhea || !hmtx) return bail(set_error(FR_E_INVALID, "missing hhea/hmtx table"));
    if (!f->has(hhea, 36)) return bail(set_error(FR_E_INVALID, "truncated hhea"));
    f->num_long_hor_metrics = f->u16(hhea + 34);
    f->pos_hmtx = hmtx;
    f->glyphs.resize(f->num_glyphs);
    // cmap: Font.zig:87-121
    if (!f->has(cmap, 4)) return bail(set_error(FR_E_INVALID, "truncated cmap"));
    const uint16_t n_sub = f->u16(cmap + 2);
    if (n_sub == 0 || !f->has(cmap + 4, 8u * n_sub)) return bail(set_error(FR_E_INVALID, "truncated cmap"));
    auto enc_at = [&](uint16_t i) { const size_t e = cmap + 4 + 8u * i; return Enc{f->u16(e), f->u16(e + 2), f->u32(e + 4)}; };
    Enc chosen = enc_at(0);
    bool best_found = false;
    for (uint16_t i = 1; i < n_sub; ++i) {
        if (is_the_best(chosen)) { best_found = true; break; }
        const Enc next = enc_at(i);
        if (is_better_than(next, chosen)) chosen = next;
    }
    if (!best_found && !is_unicode(chosen)) return bail(set_error(FR_E_UNSUPPORTED, "no unicode cmap subtable"));
    f->pos_cmap_sub = cmap + chosen.offset;
    if (!f->has(f->pos_cmap_sub, 2)) return bail(set_error(FR_E_INVALID, "truncated cmap subtable"));
    f->cmap_format = f->u16(f->pos_cmap_sub);
    if (f->cmap_format != 4 && f->cmap_format != 12)
        return bail(set_error(FR_E_UNSUPPORTED, "cmap subtable format %d (the reference supports 4 and 12)", f->cmap_format));
    *out = f;
    return FR_OK;
}

void fr_font_close(fr_font *font) { delete font; }

int fr_font_info(const fr_font *font, uint16_t *units_per_em, uint16_t *num_glyphs, int *y0_baseline)
{
    if (!font) return set_error(FR_E_INVALID, "font is NULL");
    if (units_per_em) *units_per_em = font->units_per_em;
    if (num_glyphs) *num_glyphs = font->num_glyphs;
    if (y0_baseline) *y0_baseline = font->y0_baseline ? 1 : 0;
    return FR_OK;
}

// Font.getGlyph's second result (Font.zig:161-169): advance_widths[glyph_index] as loadAdvanceWidths fills it
// (Font.zig:123-139) — for the first num_of_long_hor_metrics glyphs the advance of their LongHorMetric read as i16
// (:129), for the others the i16 entries that follow the long metrics (:132).  (Those trailing entries are the
// left-side bearings in the TrueType layout; the reference hands them out as advances — kept as it is, cited.)
int fr_font_glyph_advance(const fr_font *font, uint16_t glyph_index, int16_t *advance_width)
{
    if (!font || !advance_width) return set_error(FR_E_INVALID, "fr_font_glyph_advance: NULL argument");
    const fr_font &f = *font;
    if (glyph_index >= f.num_glyphs) return set_error(FR_E_INVALID, "glyph index %u of %u", (unsigned)glyph_index, (unsigned)f.num_glyphs);
    const size_t n_long = f.num_long_hor_metrics;
    const size_t at = glyph_index < n_long ? f.pos_hmtx + 4u * glyph_index : f.pos_hmtx + 4u * n_long + 2u * (glyph_index - n_long);
    if (!f.has(at, 2)) return set_error(FR_E_INVALID, "truncated hmtx");
    *advance_width = f.i16(at);
    return FR_OK;
}

int fr_font_char_to_glyph(const fr_font *font, uint32_t ch, uint16_t *glyph_index)
{
    if (!font || !glyph_index) return set_error(FR_E_INVALID, "fr_font_char_to_glyph: NULL argument");
    const fr_font &f = *font;
    const size_t s = f.pos_cmap_sub;
    *glyph_index = 0;
    if (f.cmap_format == 4) {                                                        // ttf.zig:433-449
        if (ch > 0xffff) return FR_OK;
        if (!f.has(s, 14)) return set_error(FR_E_INVALID, "truncated cmap format 4");
        const uint16_t seg = f.u16(s + 6) / 2;
        const size_t end_code = s + 14, start_code = end_code + 2u * seg + 2, id_delta = start_code + 2u * seg, id_range = id_delta + 2u * seg;
        if (!f.has(s, 16u + 8u * seg)) return set_error(FR_E_INVALID, "truncated cmap format 4");
        for (uint16_t i = 0; i < seg; ++i) {
            if (f.u16(end_code + 2u * i) >= ch) {
                const uint16_t st = f.u16(start_code + 2u * i);
                if (st > ch) return FR_OK;
                const uint16_t ro = f.u16(id_range + 2u * i), dl = f.u16(id_delta + 2u * i);
                if (ro != 0) {
                    const size_t a = id_range + 2u * i + ro + 2u * (ch - st);          // &idRangeOffset[i] + ro + 2*(c - start)
                    if (!f.has(a, 2)) return set_error(FR_E_INVALID, "cmap format 4: glyph index array overrun");
                    *glyph_index = (uint16_t)(dl + f.u16(a));                         // :445 (the reference adds id_delta unconditionally)
                } else {
                    *glyph_index = (uint16_t)(dl + ch);                               // :447
                }
                return FR_OK;
            }
        }
        return FR_OK;
    }
    // format 12: ttf.zig:608-614 via the range mappings (gaps map to glyph 0)
    if (!f.has(s, 16)) return set_error(FR_E_INVALID, "truncated cmap format 12");
    const uint32_t n = f.u32(s + 12);
    if (!f.has(s + 16, 12ull * n)) return set_error(FR_E_INVALID, "truncated cmap format 12");
    for (uint32_t i = 0; i < n; ++i) {
        const size_t g = s + 16 + 12ull * i;
        if (f.u32(g + 4) >= ch) {
            if (f.u32(g) <= ch) *glyph_index = (uint16_t)(f.u32(g + 8) + (ch - f.u32(g)));
            return FR_OK;
        }
    }
    return FR_OK;
}

int fr_font_glyph_measure(fr_font *font, uint16_t glyph_index, uint32_t *n_contours, uint32_t *n_points, int16_t box[4])
{
    int rc = ensure_loaded(font, glyph_index);
    if (rc) return rc;
    const GlyphData &g = font->glyphs[glyph_index];
    if (n_contours) *n_contours = (uint32_t)g.cstart.size() - 1;
    if (n_points) *n_points = (uint32_t)(g.pts.size() / 2);
    if (box) memcpy(box, g.box, sizeof g.box);
    return FR_OK;
}

int fr_font_glyph_fill(fr_font *font, uint16_t glyph_index, int16_t *points_xy, uint32_t *contour_start)
{
    int rc = ensure_loaded(font, glyph_index);
    if (rc) return rc;
    const GlyphData &g = font->glyphs[glyph_index];
    if (!g.pts.empty()) {
        if (!points_xy) return set_error(FR_E_INVALID, "points_xy is NULL");
        memcpy(points_xy, g.pts.data(), g.pts.size() * sizeof(int16_t));
    }
    if (!contour_start) return set_error(FR_E_INVALID, "contour_start is NULL");
    memcpy(contour_start, g.cstart.data(), g.cstart.size() * sizeof(uint32_t));
    return FR_OK;
}

}  // extern "C"
