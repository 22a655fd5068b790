#!/usr/bin/env python3
"""Re-emits tests/golden/ascii_glyphs.npz (the committed fixture: 190 real glyphs, expanded contours) as the flat
little-endian binary tools/zig_mint/mint_vectors.zig reads — tests/golden/ascii_glyphs.bin:

    "FRFX1\\0\\0\\0" | u32 n_glyphs | per glyph: u16 units_per_em, i16 box[4] (x_min, y_min, x_max, y_max), u32 n_contours,
                                                 per contour: u32 n_points, n_points x (i16 x, i16 y)

Data only (inputs); run from the repo root:  python tools/zig_mint/export_fixture.py [out path]"""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    import fixtures
    asc = fixtures.load_ascii()
    out = bytearray(b"FRFX1\0\0\0")
    out += struct.pack("<I", len(asc))
    for i in range(len(asc)):
        g = asc.glyph(i)
        out += struct.pack("<H4hI", int(asc.g_upm[i]), g.box.x_min, g.box.y_min, g.box.x_max, g.box.y_max, len(g.contours))
        for c in g.contours:
            pts = c.points.astype("<i2")
            out += struct.pack("<I", len(pts)) + pts.tobytes()
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "ascii_glyphs.bin")
    with open(path, "wb") as f:
        f.write(out)
    print(f"{path}: {len(asc)} glyphs, {len(out)} bytes")


if __name__ == "__main__":
    main()
