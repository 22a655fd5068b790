"""Atlas cell layout: build-defined fixed grids on the reference's sample formula
(SURVEY §8a1).  Each glyph gets a cell x cell window whose pixel (0,0) is the
glyph's own renderGlyph origin — min_x = floor(box.x_min*scale), max_y =
ceil(box.y_max*scale), in binary32 as /root/reference/src/tools/render_glyph.zig:13-17
— so a cell shows the same samples renderGlyph would, clipped / padded to the cell."""
from __future__ import annotations

import numpy as np

from .glyph import GlyphSet


def cell_jobs(gs: GlyphSet, cell: int, font_size: int, units_per_em, cols: int, *, first_glyph: int = 0,
              n_glyphs: int | None = None, page_rows: int | None = None) -> np.ndarray:
    """fr_job table: glyph i -> cell (i % cols, i // cols) of a cols-wide atlas."""
    n = len(gs) - first_glyph if n_glyphs is None else n_glyphs
    dt = np.dtype([("glyph", "<u4"), ("min_x", "<i4"), ("max_y", "<i4"), ("w", "<u4"), ("h", "<u4"),
                   ("out_x", "<u4"), ("out_y", "<u4"), ("scale", "<f4")])
    jobs = np.zeros(n, dt)
    upm = np.broadcast_to(np.asarray(units_per_em, np.float32), (len(gs),))[first_glyph:first_glyph + n]
    scale = np.float32(font_size) / upm.astype(np.float32)                     # :13
    box = gs.boxes[first_glyph:first_glyph + n].astype(np.float32)
    idx = np.arange(n)
    jobs["glyph"] = idx + first_glyph
    jobs["min_x"] = np.floor(box[:, 0] * scale).astype(np.int32)               # :16
    jobs["max_y"] = np.ceil(box[:, 3] * scale).astype(np.int32)                # :17
    jobs["w"] = cell
    jobs["h"] = cell
    jobs["out_x"] = (idx % cols) * cell
    jobs["out_y"] = (idx // cols) * cell
    jobs["scale"] = scale
    return jobs


def atlas_shape(n_glyphs: int, cell: int, cols: int):
    rows = (n_glyphs + cols - 1) // cols
    return rows * cell, cols * cell


def atlas_pages(n_glyphs: int, cell: int, page: int = 2048):
    """cell grid on page x page atlas pages (BASELINE configs[1]: 128-cells on one 2048^2 page):
    -> list of (first_glyph, n_glyphs_on_page); cols = page // cell for every page"""
    per = (page // cell) ** 2
    return [(s, min(per, n_glyphs - s)) for s in range(0, n_glyphs, per)]
