// fr_host.hpp — C++ host-side mirror of the reference's interface for this path, sitting on
// the C ABI (include/fr_raster.h).  The reference is compiled code (Zig) and no Zig toolchain
// exists in the build image, so this header is the compiled-language host a maintainer can
// read next to the Zig: same names, argument meaning and error behaviour as
//   /root/reference/src/font/Glyph.zig:11-24          (Glyph, Box, Contour)
//   /root/reference/src/font/Font.zig:25-29           (Font::Information)
//   /root/reference/src/tools/Image.zig:44-130        (Image::Gray, Image::Winding)
//   /root/reference/src/tools/render_glyph.zig:11,160 (renderGlyph, windingInGlyph)
// Nothing is computed here: every call goes to libfr_raster.so (HIP, gfx950).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fr_raster.h"

namespace fr_host {

struct Point { int16_t x, y; };                       // geometry.zig:7-11 (extern struct)
struct Box { int16_t x_min = 0, y_min = 0, x_max = 0, y_max = 0; };   // Glyph.zig:15-20
struct Contour { std::vector<Point> points; };        // Glyph.zig:22-24: even on-curve, odd control, last == first
struct Glyph {                                        // Glyph.zig:11-12
    Box box;
    std::vector<Contour> contours;
    static Glyph initEmpty() { return {}; }           // Glyph.zig:77-82
};
struct FontInformation { uint16_t units_per_em; bool y0_baseline = true; };   // Font.zig:25-29

struct RasterFailed : std::runtime_error {            // the Zig shim's error.RasterFailed
    int code;
    RasterFailed(int c, const char *m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) { if (rc != FR_OK) throw RasterFailed(rc, fr_last_error()); }

namespace Image {
struct Gray {                                         // Image.zig:44-83
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> data;                        // row-major, y*width + x
    static Gray init(uint32_t w, uint32_t h) { Gray g; g.width = w; g.height = h; g.data.assign((size_t)w * h, 0); return g; }
    void getRGBLinear(size_t index, uint8_t rgb[3]) const { rgb[0] = rgb[1] = rgb[2] = data[index]; }   // :78-82
};
struct Winding {                                      // Image.zig:85-130
    uint32_t width = 0, height = 0;
    std::vector<int16_t> data;
    uint8_t scaler = 50, overflow_color = 150;
    static Winding init(uint32_t w, uint32_t h, uint8_t scaler, uint8_t overflow) {
        Winding g; g.width = w; g.height = h; g.data.assign((size_t)w * h, 0); g.scaler = scaler; g.overflow_color = overflow; return g;
    }
    void getRGBLinear(size_t index, uint8_t rgb[3]) const {                                             // :121-129
        const int16_t val = data[index];
        if (val == 0) { rgb[0] = rgb[1] = rgb[2] = 0; return; }
        uint32_t c = (uint32_t)scaler * (uint32_t)(val < 0 ? -val : val);
        if (c > 65535u) c = 65535u;                   // u16 saturating multiply
        const uint8_t color = (uint8_t)(c > 255u ? 255u : c);
        const uint8_t sub = (c == color) ? 0 : overflow_color;
        if (val > 0) { rgb[0] = sub; rgb[1] = sub; rgb[2] = color; } else { rgb[0] = color; rgb[1] = sub; rgb[2] = sub; }
    }
};
struct RGB {                                          // Image.zig:132-170
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> data;                        // row-major RGB triples
    static RGB init(uint32_t w, uint32_t h) { RGB g; g.width = w; g.height = h; g.data.assign((size_t)w * h * 3, 0); return g; }
    void getRGBLinear(size_t index, uint8_t rgb[3]) const { rgb[0] = data[3 * index]; rgb[1] = data[3 * index + 1]; rgb[2] = data[3 * index + 2]; }
};
struct GlyphDebug {                                   // Image.zig:173-241
    RGB rgb;
    uint8_t winding_scale = 50, overflow_color = 150;
};
}  // namespace Image

class Context {
public:
    explicit Context(int device = 0, void *stream = nullptr) { check(fr_ctx_create(device, stream, &h_)); }
    ~Context() { fr_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    fr_ctx *get() const { return h_; }
private:
    fr_ctx *h_ = nullptr;
};

// flat view the C ABI takes: points back to back + contour offsets
struct Flat {
    std::vector<int16_t> pts;
    std::vector<uint32_t> cstart{0};
    explicit Flat(const Glyph &g) {
        for (const Contour &c : g.contours) {
            for (const Point &p : c.points) { pts.push_back(p.x); pts.push_back(p.y); }
            cstart.push_back((uint32_t)(pts.size() / 2));
        }
        if (pts.empty()) pts.assign(2, 0);
    }
    uint32_t n_contours() const { return (uint32_t)cstart.size() - 1; }
};

// render_glyph.zig:11 — pub fn renderGlyph(glyph, font_info, font_size) !Image.Gray
inline Image::Gray renderGlyph(Context &ctx, const Glyph &glyph, FontInformation font_info, uint16_t font_size)
{
    const int16_t box[4] = {glyph.box.x_min, glyph.box.y_min, glyph.box.x_max, glyph.box.y_max};
    int16_t mn[2], mx[2];
    uint16_t w, h;
    check(fr_render_glyph_dims(box, font_info.units_per_em, font_size, mn, mx, &w, &h, nullptr));
    Image::Gray im = Image::Gray::init(w, h);                                   // :22
    Flat f(glyph);
    check(fr_render_glyph(ctx.get(), f.pts.data(), f.cstart.data(), f.n_contours(), box, font_info.units_per_em,
                          font_size, FR_GRAY_DEBUG, im.data.data()));
    return im;
}

// same grid, raw winding numbers (glyphWindingAt, :35-73) into an Image::Winding
inline Image::Winding renderGlyphWinding(Context &ctx, const Glyph &glyph, FontInformation font_info, uint16_t font_size)
{
    const int16_t box[4] = {glyph.box.x_min, glyph.box.y_min, glyph.box.x_max, glyph.box.y_max};
    int16_t mn[2], mx[2];
    uint16_t w, h;
    check(fr_render_glyph_dims(box, font_info.units_per_em, font_size, mn, mx, &w, &h, nullptr));
    Image::Winding im = Image::Winding::init(w, h, 50, 150);
    Flat f(glyph);
    check(fr_render_glyph(ctx.get(), f.pts.data(), f.cstart.data(), f.n_contours(), box, font_info.units_per_em,
                          font_size, FR_WINDING_I16, im.data.data()));
    return im;
}

// render_glyph.zig:160 — windingInGlyph(glyph, glyph_info, point) i16 (GlyphInfo is recomputed on the device)
inline int16_t windingInGlyph(Context &ctx, const Glyph &glyph, Point p)
{
    Flat f(glyph);
    const int16_t q[2] = {p.x, p.y};
    int16_t out = 0;
    check(fr_winding_in_glyph(ctx.get(), f.pts.data(), f.cstart.data(), f.n_contours(), q, 1, &out));
    return out;
}

// Image.zig:220 — Image.GlyphDebug.render(glyph, winding_scale): the exact-integer lattice coloured, points marked
inline Image::GlyphDebug glyphDebugRender(Context &ctx, const Glyph &glyph, uint8_t winding_scale)
{
    const int16_t box[4] = {glyph.box.x_min, glyph.box.y_min, glyph.box.x_max, glyph.box.y_max};
    Image::GlyphDebug im;
    im.winding_scale = winding_scale;
    im.rgb = Image::RGB::init((uint32_t)(box[2] - box[0] + 3), (uint32_t)(box[3] - box[1] + 3));     // :183
    Flat f(glyph);
    check(fr_glyph_debug_render(ctx.get(), f.pts.data(), f.cstart.data(), f.n_contours(), box, winding_scale, im.rgb.data.data()));
    return im;
}

// a batch of glyphs -> one atlas (build-defined cell grid, fr_atlas_layout) of n x n-sample coverage
// optional multi-GPU assembly (include/fr_raster.h): rank r rendered its band at atlas_dev + r * band_bytes; comm is the
// host's ncclComm_t
inline void allgatherBands(Context &ctx, void *nccl_comm, void *atlas_dev, size_t band_bytes)
{
    check(fr_allgather_bands(ctx.get(), nccl_comm, atlas_dev, band_bytes));
}

inline Image::Gray renderAtlas(Context &ctx, const std::vector<Glyph> &glyphs, FontInformation font_info, uint16_t font_size,
                               uint32_t cell, uint32_t cols, int samples_per_axis)
{
    std::vector<int16_t> pts, boxes;
    std::vector<uint32_t> cstart{0}, gstart{0};
    for (const Glyph &g : glyphs) {
        for (const Contour &c : g.contours) {
            for (const Point &p : c.points) { pts.push_back(p.x); pts.push_back(p.y); }
            cstart.push_back((uint32_t)(pts.size() / 2));
        }
        gstart.push_back((uint32_t)cstart.size() - 1);
        boxes.insert(boxes.end(), {g.box.x_min, g.box.y_min, g.box.x_max, g.box.y_max});
    }
    if (pts.empty()) pts.assign(2, 0);
    const uint32_t n = (uint32_t)glyphs.size();
    std::vector<fr_job> jobs(n ? n : 1);
    const uint16_t upm = font_info.units_per_em;
    check(fr_atlas_layout(boxes.data(), n, 0, &upm, 1, font_size, cell, cols, 0, jobs.data(), nullptr, nullptr));
    fr_glyphset *gs = nullptr;
    check(fr_glyphset_create(ctx.get(), pts.data(), cstart.data(), (uint32_t)cstart.size() - 1, gstart.data(), n, &gs));
    const uint32_t rows = (n + cols - 1) / cols;
    Image::Gray im = Image::Gray::init(cols * cell, rows * cell);
    fr_raster_params prm{FR_COVERAGE_U8, samples_per_axis, FR_SAMPLE_CENTER, 0};
    const int rc = fr_render_batch(ctx.get(), gs, jobs.data(), n, &prm, im.data.data(), (size_t)cols * cell, (size_t)rows * cell);
    fr_glyphset_destroy(gs);
    check(rc);
    return im;
}

}  // namespace fr_host
