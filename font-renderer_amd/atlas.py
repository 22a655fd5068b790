"""Atlas cell layout: build-defined fixed grids on the reference's sample formula
(SURVEY §8a1).  Each glyph gets a cell x cell window whose pixel (0,0) is the
glyph's own renderGlyph origin — min_x = floor(box.x_min*scale), max_y =
ceil(box.y_max*scale), in binary32 as /root/reference/src/tools/render_glyph.zig:13-17
— so a cell shows the same samples renderGlyph would, clipped / padded to the cell.
The arithmetic lives behind the C ABI (fr_atlas_layout, host-side C++): this module only
marshals arrays, so a Zig host gets the very same job table."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .glyph import GlyphSet

JOB_DTYPE = np.dtype([("glyph", "<u4"), ("min_x", "<i4"), ("max_y", "<i4"), ("w", "<u4"), ("h", "<u4"),
                      ("out_x", "<u4"), ("out_y", "<u4"), ("scale", "<f4")])


def cell_jobs(gs: GlyphSet, cell: int, font_size: int, units_per_em, cols: int, *, first_glyph: int = 0,
              n_glyphs: int | None = None, page_rows: int | None = None, return_pages: bool = False):
    """fr_job table: glyph i -> cell (i % cols, i // cols) of a cols-wide atlas; page_rows cuts the grid into
    pages of that many cell rows (out_y restarts per page; return_pages=True also returns the page index of
    every job and the page count)."""
    n = len(gs) - first_glyph if n_glyphs is None else n_glyphs
    jobs = np.zeros(n, JOB_DTYPE)
    upm = np.broadcast_to(np.asarray(units_per_em), (len(gs),))[first_glyph:first_glyph + n]
    upm = np.ascontiguousarray(upm, np.uint16) if n else np.ones(1, np.uint16)
    boxes = np.ascontiguousarray(gs.boxes[first_glyph:first_glyph + n], np.int16)
    pages = np.zeros(max(n, 1), np.uint32)
    n_pages = C.c_uint32()
    L.check(L.load_library().fr_atlas_layout(L.ptr(boxes), n, first_glyph, L.ptr(upm), len(upm) if n else 1, font_size, cell, cols,
                                             page_rows or 0, L.ptr(jobs), L.ptr(pages), C.byref(n_pages)))
    if return_pages:
        return jobs, pages[:n], n_pages.value
    return jobs


def glyph_dims_jobs(gs: GlyphSet, font_size: int, units_per_em, atlas_w: int, *, align: int = 1, first_glyph: int = 0,
                    n_glyphs: int | None = None):
    """fr_job table of the reference's own product in batch: glyph i at exactly renderGlyph's image size
    (render_glyph.zig:13-19), shelf-packed into an atlas_w-wide atlas (fr_atlas_layout_glyph_dims) -> (jobs, atlas_h)"""
    n = len(gs) - first_glyph if n_glyphs is None else n_glyphs
    jobs = np.zeros(n, JOB_DTYPE)
    upm = np.broadcast_to(np.asarray(units_per_em), (len(gs),))[first_glyph:first_glyph + n]
    upm = np.ascontiguousarray(upm, np.uint16) if n else np.ones(1, np.uint16)
    boxes = np.ascontiguousarray(gs.boxes[first_glyph:first_glyph + n], np.int16)
    h = C.c_uint32()
    L.check(L.load_library().fr_atlas_layout_glyph_dims(L.ptr(boxes), n, first_glyph, L.ptr(upm), len(upm) if n else 1, font_size,
                                                        atlas_w, align, L.ptr(jobs), C.byref(h)))
    return jobs, h.value


def atlas_shape(n_glyphs: int, cell: int, cols: int):
    rows = (n_glyphs + cols - 1) // cols
    return rows * cell, cols * cell


def atlas_pages(n_glyphs: int, cell: int, page: int = 2048):
    """cell grid on page x page atlas pages (BASELINE configs[1]: 128-cells on one 2048^2 page):
    -> list of (first_glyph, n_glyphs_on_page); cols = page // cell for every page"""
    per = (page // cell) ** 2
    return [(s, min(per, n_glyphs - s)) for s in range(0, n_glyphs, per)]
