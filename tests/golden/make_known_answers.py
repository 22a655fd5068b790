#!/usr/bin/env python3
"""Mint tests/golden/known_answers.json from the C oracle (SHA-256 prefixes of images).
These are the build's own known answers — the reference holds none."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]


def _h(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def compute(oracle, ascii_set):
    import oracle_lib as O
    from font_renderer_amd.atlas import cell_jobs
    out = {}
    for font, ch, size in [("STIX", "A", 64), ("STIX", "&", 48), ("DejaVu", "A", 64), ("DejaVu", "g", 57)]:
        i = ascii_set.find(font, ch)
        g, upm = ascii_set.glyph(i), int(ascii_set.g_upm[i])
        out[f"gray/{font}/{ch}/{size}"] = _h(oracle.render_glyph(g, upm, size))
        mn, mx, w, h, s = oracle.render_glyph_dims(g.box.as_array(), upm, size)
        out[f"cov4c/{font}/{ch}/{size}"] = _h(oracle.render_cell(g, mn[0], mx[1], w, h, s, O.COVERAGE_U8, 4, True))
        out[f"lattice/{font}/{ch}"] = _h(oracle.winding_lattice(g).astype("<i2"))
    # a small atlas: 16 STIX glyphs, 32x32 cells, 2x2 centre samples
    jobs = cell_jobs(ascii_set.gs, 32, 30, ascii_set.g_upm, 4, first_glyph=33, n_glyphs=16)
    atlas = np.zeros((128, 128), np.uint8)
    oracle.render_batch(ascii_set.gs, jobs, O.COVERAGE_U8, atlas, 2, True)
    out["atlas/stix16/32/cov2c"] = _h(atlas)
    return out


if __name__ == "__main__":
    import fixtures
    import oracle_lib
    ka = compute(oracle_lib.Oracle(), fixtures.load_ascii())
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1, sort_keys=True)
    print(json.dumps(ka, indent=1))
