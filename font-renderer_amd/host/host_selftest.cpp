// host_selftest.cpp — exercises the C++ host mirror (fr_host.hpp) against a known answer:
// STIXGeneral 'A' at font_size 64.  Built by __graft_entry__.build();
// run on the GPU box by tests/test_gpu_parity.py::test_cpp_host_mirror.
// usage: host_selftest <points.bin: i16 pairs> <cstart.bin: u32> ; prints "w h hist(-2,-1,0,1) fnv1a(gray) winding@(420,321) fnv1a(GlyphDebug) fnv1a(atlas)"
#include <cstdio>
#include <cstdlib>
#include <map>

#include "fr_host.hpp"

template <class T> static std::vector<T> slurp(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<T> v((size_t)n / sizeof(T));
    if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) exit(2);
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    try {
        auto pts = slurp<int16_t>(argv[1]);
        auto cs = slurp<uint32_t>(argv[2]);
        fr_host::Glyph g;
        g.box = {15, 0, 707, 674};
        for (size_t c = 0; c + 1 < cs.size(); ++c) {
            fr_host::Contour ct;
            for (uint32_t i = cs[c]; i < cs[c + 1]; ++i) ct.points.push_back({pts[2 * i], pts[2 * i + 1]});
            g.contours.push_back(ct);
        }
        fr_host::Context ctx(0);
        auto gray = fr_host::renderGlyph(ctx, g, {1000}, 64);
        auto wd = fr_host::renderGlyphWinding(ctx, g, {1000}, 64);
        std::map<int, int> hist;
        for (int16_t v : wd.data) hist[v]++;
        uint64_t h = 1469598103934665603ull;
        for (uint8_t b : gray.data) { h ^= b; h *= 1099511628211ull; }
        // Image.GlyphDebug.render and a 2 x 1 atlas of 128-pixel cells (16 samples per pixel: cov4_kernel), FNV-1a of each
        auto dbg = fr_host::glyphDebugRender(ctx, g, 50);
        uint64_t hd = 1469598103934665603ull;
        for (uint8_t b : dbg.rgb.data) { hd ^= b; hd *= 1099511628211ull; }
        auto atlas = fr_host::renderAtlas(ctx, {g, g}, {1000}, 100, 128, 2, 4);
        uint64_t ha = 1469598103934665603ull;
        for (uint8_t b : atlas.data) { ha ^= b; ha *= 1099511628211ull; }
        printf("%u %u %d %d %d %d %016llx %d %016llx %016llx\n", gray.width, gray.height, hist[-2], hist[-1], hist[0], hist[1],
               (unsigned long long)h, (int)fr_host::windingInGlyph(ctx, g, {420, 321}), (unsigned long long)hd, (unsigned long long)ha);
        return 0;
    } catch (const fr_host::RasterFailed &e) {
        fprintf(stderr, "RasterFailed(%d): %s\n", e.code, e.what());
        return 1;
    }
}
