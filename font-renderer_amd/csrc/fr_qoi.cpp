// fr_qoi.cpp — QOI writer for atlases / debug images, byte-compatible with the
// reference's streaming encoder /root/reference/src/tools/qoi.zig:25-88 (saveRGB over the Image
// vtable): RGB only, sRGB colorspace tag, op order RUN -> INDEX -> DIFF -> LUMA -> RGB, run cap
// 62 (0x3D + 1), big-endian header, 8-byte end marker.  Host code (the format is sequential).
// Note the reference's quirk, kept: the "previous pixel" before the first one is {0,0,0} and the
// running table is written BEFORE the new pixel is read (:42-45), i.e. with the previous pixel.
#include "../../include/fr_raster.h"

#include <cstring>

namespace fr { int set_error(int code, const char *fmt, ...); }

namespace {

struct Px { uint8_t r, g, b; };
inline bool same(Px a, Px b) { return a.r == b.r && a.g == b.g && a.b == b.b; }
inline uint8_t hash(Px p) { return (uint8_t)((uint8_t)(p.r * 3) + (uint8_t)(p.g * 5) + (uint8_t)(p.b * 7) + (uint8_t)(255 * 11)) & 63; }   // :85-88

template <class Get>
int encode(Get get, uint32_t w, uint32_t h, uint8_t *out, size_t cap, size_t *n_out)
{
    const size_t total = (size_t)w * h;
    size_t o = 0;
    auto put = [&](uint8_t b) { if (o < cap) out[o] = b; ++o; };
    put('q'); put('o'); put('i'); put('f');                                   // :26-32
    put(w >> 24); put(w >> 16); put(w >> 8); put(w);
    put(h >> 24); put(h >> 16); put(h >> 8); put(h);
    put(3); put(0);
    Px running[64];
    memset(running, 0, sizeof running);                                        // :35
    Px prev{0, 0, 0}, curr{0, 0, 0};                                           // :37-38
    uint8_t ridx = 0;
    size_t next = 0;
    while (next < total) {                                                     // :41
        running[ridx] = curr;                                                  // :42
        prev = curr;
        curr = get(next);
        ridx = hash(curr);
        ++next;
        if (same(prev, curr)) {                                                // :48
            uint8_t run = 0;
            while (run < 0x3D && next < total && same(prev, get(next))) { ++next; ++run; }
            put(0xC0 | run);
        } else if (same(running[ridx], curr)) {                                // :61
            put(ridx);
        } else {
            uint8_t dr = (uint8_t)(curr.r - prev.r + 2), dg = (uint8_t)(curr.g - prev.g + 2), db = (uint8_t)(curr.b - prev.b + 2);   // :67
            if (dr < 4 && dg < 4 && db < 4) {
                put(0x40 | (dr << 4) | (dg << 2) | db);                        // :69
            } else {
                dr = (uint8_t)(dr + (uint8_t)(8 - dg));                        // :73
                db = (uint8_t)(db + (uint8_t)(8 - dg));
                dg = (uint8_t)(dg + 30);
                if (dr < 16 && dg < 64 && db < 16) {
                    put(0x80 | dg); put((dr << 4) | db);                       // :75
                } else {
                    put(0xFE); put(curr.r); put(curr.g); put(curr.b);          // :80
                }
            }
        }
    }
    for (int i = 0; i < 7; ++i) put(0);                                        // :82
    put(1);
    if (n_out) *n_out = o;
    if (o > cap) return fr::set_error(FR_E_INVALID, "QOI output needs %zu bytes, buffer has %zu", o, cap);
    return FR_OK;
}

}  // namespace

extern "C" {

size_t fr_qoi_bound(uint32_t width, uint32_t height) { return 14 + 8 + (size_t)width * height * 4; }

int fr_qoi_encode_rgb(const uint8_t *rgb, uint32_t width, uint32_t height, uint8_t *out, size_t cap, size_t *n_out)
{
    if (!rgb || !out) return fr::set_error(FR_E_INVALID, "fr_qoi_encode_rgb: NULL argument");
    return encode([&](size_t i) { return Px{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]}; }, width, height, out, cap, n_out);
}

// Image.Gray through the Image vtable: getRGBLinear = {v, v, v} (Image.zig:78-82)
int fr_qoi_encode_gray(const uint8_t *gray, uint32_t width, uint32_t height, size_t stride, uint8_t *out, size_t cap, size_t *n_out)
{
    if (!gray || !out) return fr::set_error(FR_E_INVALID, "fr_qoi_encode_gray: NULL argument");
    if (stride < width) return fr::set_error(FR_E_INVALID, "stride < width");
    return encode([&](size_t i) { const uint8_t v = gray[(i / width) * stride + (i % width)]; return Px{v, v, v}; }, width, height, out, cap, n_out);
}

}  // extern "C"
