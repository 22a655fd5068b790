#!/usr/bin/env python3
"""Mint the real-outline fixture tests/golden/ascii_glyphs.npz.

Run in the build container only (needs fontTools + matplotlib's bundled fonts);
the GPU box and the tests only read the committed .npz (numpy.load, no pickle).

What is stored, per glyph (95 printable ASCII x 2 unhinted fonts the reference
itself would accept — Glyph.zig:85 panics on glyphs that carry instructions):
  * the RAW TrueType simple-glyph data (absolute coords, on-curve flags,
    endPtsOfContours, bbox) exactly as ttf.SimpleGlyph would hand it to
    Glyph.initTTFSimple (/root/reference/src/font/Glyph.zig:84-106);
  * the EXPANDED contour points computed here by an independent Python
    restatement of Contour.initTTF (Glyph.zig:43-74, truncating midpoint
    geometry.zig:12-17) — the C oracle's or_contour_init_ttf must reproduce them.
The fonts are data inputs (DejaVu: Bitstream-Vera licence, STIX: OFL); nothing of
the reference is copied or executed.
"""
import os
import sys

import numpy as np
from fontTools.ttLib import TTFont

FONT_DIR = "/usr/local/lib/python3.10/dist-packages/matplotlib/mpl-data/fonts/ttf"
FONTS = ["STIXGeneral.ttf", "DejaVuSerif-Italic.ttf"]


def div_trunc(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def expand_contour(coords, on, start, end):
    """Contour.initTTF, Glyph.zig:43-74."""
    buf = {}
    prev_on = on[end]
    prev = coords[end]
    nxt = 1 if prev_on else 0
    for i in range(start, end + 1):
        cur_on, cur = on[i], coords[i]
        if prev_on == cur_on:
            buf[nxt] = (div_trunc(prev[0] + cur[0], 2), div_trunc(prev[1] + cur[1], 2))
            nxt += 1
        buf[nxt] = cur
        nxt += 1
        prev_on, prev = cur_on, cur
    if prev_on:
        buf[0] = buf[nxt - 1]
    else:
        buf[nxt] = buf[0]
        nxt += 1
    return [buf[i] for i in range(nxt)]


def main(out_path):
    g_font, g_char, g_box, g_upm = [], [], [], []
    raw_coords, raw_on, raw_end = [], [], []
    g_coord_start, g_end_start = [0], [0]
    exp_pts, exp_cstart, g_cont_start = [], [0], [0]
    skipped = []
    for fi, fname in enumerate(FONTS):
        font = TTFont(os.path.join(FONT_DIR, fname))
        upm = font["head"].unitsPerEm
        cmap = font.getBestCmap()
        glyf = font["glyf"]
        for ch in range(0x20, 0x7F):
            g = glyf[cmap[ch]]
            if g.numberOfContours < 0:
                skipped.append((fname, ch, "composite"))
                continue
            ncont = max(g.numberOfContours, 0)
            if ncont and len(g.program.getBytecode()) > 0:
                skipped.append((fname, ch, "hinted"))   # reference panics, Glyph.zig:85
                continue
            coords = [tuple(c) for c in g.coordinates] if ncont else []
            on = [bool(f & 1) for f in g.flags] if ncont else []
            ends = list(g.endPtsOfContours) if ncont else []
            box = (g.xMin, g.yMin, g.xMax, g.yMax) if ncont else (0, 0, 0, 0)
            g_font.append(fi); g_char.append(ch); g_box.append(box); g_upm.append(upm)
            raw_coords += coords; raw_on += on; raw_end += ends
            g_coord_start.append(len(raw_coords)); g_end_start.append(len(raw_end))
            s = 0
            for e in ends:
                pts = expand_contour(coords, on, s, e)
                exp_pts += pts
                exp_cstart.append(len(exp_pts))
                s = e + 1
            g_cont_start.append(len(exp_cstart) - 1)
    np.savez_compressed(
        out_path,
        font_names=np.array(FONTS),
        g_font=np.array(g_font, np.uint8), g_char=np.array(g_char, np.uint32),
        g_box=np.array(g_box, np.int16).reshape(-1, 4), g_upm=np.array(g_upm, np.uint16),
        raw_coords=np.array(raw_coords, np.int16).reshape(-1, 2),
        raw_on=np.array(raw_on, np.uint8), raw_end=np.array(raw_end, np.uint16),
        g_coord_start=np.array(g_coord_start, np.uint32), g_end_start=np.array(g_end_start, np.uint32),
        exp_pts=np.array(exp_pts, np.int16).reshape(-1, 2),
        exp_cstart=np.array(exp_cstart, np.uint32), g_cont_start=np.array(g_cont_start, np.uint32),
    )
    print(f"wrote {out_path}: {len(g_char)} glyphs, {len(exp_pts)} expanded points, skipped={skipped}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "ascii_glyphs.npz"))
