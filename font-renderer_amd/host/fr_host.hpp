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

}  // namespace fr_host
