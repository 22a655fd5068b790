"""Glyph model — mirror of /root/reference/src/font/Glyph.zig:11-24 and
Font.Information (/root/reference/src/font/Font.zig:25-29)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np


@dataclass
class Box:                      # Glyph.zig:15-20
    x_min: int = 0
    y_min: int = 0
    x_max: int = 0
    y_max: int = 0

    def as_array(self) -> np.ndarray:
        return np.array([self.x_min, self.y_min, self.x_max, self.y_max], np.int16)


@dataclass
class Contour:                  # Glyph.zig:22-24
    """points: (n, 2) int16; even index on-curve, odd index control, last == first."""
    points: np.ndarray

    def __post_init__(self):
        self.points = np.ascontiguousarray(self.points, np.int16).reshape(-1, 2)

    @property
    def curve_count(self) -> int:   # render_glyph.zig:38
        return len(self.points) // 2


@dataclass
class Glyph:                    # Glyph.zig:11-12
    box: Box = field(default_factory=Box)
    contours: List[Contour] = field(default_factory=list)

    @staticmethod
    def initEmpty() -> "Glyph":  # Glyph.zig:77-82
        return Glyph(Box(), [])

    @property
    def curve_count(self) -> int:
        return sum(c.curve_count for c in self.contours)


@dataclass
class FontInformation:          # Font.zig:25-29
    units_per_em: int
    y0_baseline: bool = True
    loca_format: int = 0


class GlyphSet:
    """Glyph[] flattened the way include/fr_raster.h takes it (CSR):
    points_xy (P,2) i16, contour_start (C+1) u32 in points, glyph_start (G+1) u32 in contours."""

    def __init__(self, glyphs: Sequence[Glyph]):
        pts, cstart, gstart, boxes = [], [0], [0], []
        n = 0
        for g in glyphs:
            for c in g.contours:
                pts.append(c.points)
                n += len(c.points)
                cstart.append(n)
            gstart.append(len(cstart) - 1)
            boxes.append(g.box.as_array())
        self.points_xy = (np.concatenate(pts) if pts else np.zeros((0, 2), np.int16)).astype(np.int16)
        self.contour_start = np.array(cstart, np.uint32)
        self.glyph_start = np.array(gstart, np.uint32)
        self.boxes = np.array(boxes, np.int16).reshape(-1, 4)

    @classmethod
    def from_arrays(cls, points_xy, contour_start, glyph_start, boxes) -> "GlyphSet":
        self = cls.__new__(cls)
        self.points_xy = np.ascontiguousarray(points_xy, np.int16).reshape(-1, 2)
        self.contour_start = np.ascontiguousarray(contour_start, np.uint32)
        self.glyph_start = np.ascontiguousarray(glyph_start, np.uint32)
        self.boxes = np.ascontiguousarray(boxes, np.int16).reshape(-1, 4)
        return self

    def __len__(self) -> int:
        return len(self.glyph_start) - 1

    @property
    def n_contours(self) -> int:
        return len(self.contour_start) - 1

    def glyph(self, i: int) -> Glyph:
        c0, c1 = int(self.glyph_start[i]), int(self.glyph_start[i + 1])
        cs = [Contour(self.points_xy[int(self.contour_start[c]):int(self.contour_start[c + 1])]) for c in range(c0, c1)]
        b = self.boxes[i]
        return Glyph(Box(int(b[0]), int(b[1]), int(b[2]), int(b[3])), cs)

    def segments_per_glyph(self) -> np.ndarray:
        clen = np.diff(self.contour_start.astype(np.int64)) // 2
        cum = np.concatenate([[0], np.cumsum(clen)])
        return cum[self.glyph_start[1:].astype(np.int64)] - cum[self.glyph_start[:-1].astype(np.int64)]

    def subset(self, lo: int, hi: int) -> "GlyphSet":
        """glyphs [lo, hi) as an independent set (multi-GPU sharding)."""
        c0, c1 = int(self.glyph_start[lo]), int(self.glyph_start[hi])
        p0, p1 = int(self.contour_start[c0]), int(self.contour_start[c1])
        return GlyphSet.from_arrays(self.points_xy[p0:p1], self.contour_start[c0:c1 + 1] - np.uint32(p0),
                                    self.glyph_start[lo:hi + 1] - np.uint32(c0), self.boxes[lo:hi])
