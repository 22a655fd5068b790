#!/usr/bin/env python3
"""Timing of the exact-integer path — whole calls at the Python mirror, synchronous:
upload of the glyph, GlyphInfo.init + windingInGlyph over the lattice on the GPU, download.
  * winding_lattice / glyph_debug_render: Image.GlyphDebug.render's 1-px-per-font-unit lattice of STIX 'A' (695 x 677)
  * exact_coverage: K = 8 refined lattice, 4 x 4 points per pixel, a 256 x 256-pixel window
Prints one JSON line (profiles/r02/exact_lattice.json)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import fixtures  # noqa: E402
import font_renderer_amd as fr  # noqa: E402
from font_renderer_amd import render_glyph as rg  # noqa: E402

asc = fixtures.load_ascii()
g = asc.glyph(asc.find("STIX", "A"))
ctx = fr.Context(0)


def timed(fn, reps):
    for _ in range(20):
        out = fn()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t) / reps * 1e3, out


ms_l, lat = timed(lambda: rg.winding_lattice(g, ctx=ctx), 200)
ms_d, dbg = timed(lambda: rg.glyph_debug_render(g, 50, ctx=ctx), 200)
ms_c, cov = timed(lambda: rg.exact_coverage(g, 8, 0, 5600, 256, 256, 4, ctx=ctx), 100)
h, w = lat.shape
print(json.dumps({
    "glyph": "STIX 'A' (25 curves)", "build_id": rg.build_id(),
    "winding_lattice": {"points": [w, h], "ms_per_call": round(ms_l, 3), "Mpoints_per_s": round(w * h / ms_l / 1e3, 1)},
    "glyph_debug_render": {"image": [dbg.width, dbg.height], "ms_per_call": round(ms_d, 3), "Mpixel_per_s": round(dbg.width * dbg.height / ms_d / 1e3, 1)},
    "exact_coverage_K8_16pts": {"pixels": [256, 256], "lattice_points": 256 * 256 * 16, "ms_per_call": round(ms_c, 3),
                                "Mpoints_per_s": round(256 * 256 * 16 / ms_c / 1e3, 1)},
    "note": "whole synchronous calls (upload + kernels + download), one lane per lattice point, i64 / 128-bit predicates",
}))
