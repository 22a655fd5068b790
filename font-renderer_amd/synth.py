"""Synthetic glyph outlines: SplitMix64-seeded, star-shaped closed
quadratic splines in the Glyph.zig contour layout (even = on-curve, odd = control,
last == first; straight edges carry the TRUNCATED midpoint, geometry.zig:12-17).

Used by bench.py and the parity tests; the same arrays feed the HIP path and the
oracle.  No font file exists locally for CJK-scale sets, so configs 3-5 use these."""
from __future__ import annotations

import numpy as np

from .glyph import GlyphSet

EM = 2048
LO, HI = 64, 1984
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n outputs of SplitMix64 started at `seed` (counter form), uint64"""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.arange(1, n + 1, dtype=np.uint64) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform(bits: np.ndarray) -> np.ndarray:
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _div_trunc2(v: np.ndarray) -> np.ndarray:
    return np.where(v >= 0, v // 2, -((-v) // 2))


def synth_glyph(index: int, n_segments: int):
    """-> (list of (len,2) int16 contour arrays, box[4])"""
    S = int(n_segments)
    assert S >= 3
    rnd = splitmix64(0xF0175EED ^ index, 8 + 4 * S + 16)
    u = _uniform(rnd)
    C = 1 + int(rnd[0] % np.uint64(3))
    while C > 1 and S // C < 4:
        C -= 1
    # segments per contour: the outer contour takes the larger share
    counts = [S // C] * C
    counts[0] += S - sum(counts)
    cx0 = EM / 2 + (u[1] * 2 - 1) * 64
    cy0 = EM / 2 + (u[2] * 2 - 1) * 64
    base_r = 760.0 * (0.85 + 0.15 * u[3])
    contours, o = [], 8
    for k, Sk in enumerate(counts):
        uu = u[o:o + 4 * Sk]; o += 4 * Sk
        i = np.arange(Sk)
        ang = 2 * np.pi * (i + uu[0:Sk] * 0.5) / Sk
        rk = base_r * (0.6 ** k)
        rad = rk * (1 + 0.25 * (uu[Sk:2 * Sk] * 2 - 1))
        if k % 2 == 0:
            ang = -ang                     # clockwise in y-up font units: TrueType outer contour
        on = np.stack([cx0 + rad * np.cos(ang), cy0 + rad * np.sin(ang)], 1)
        on = np.clip(np.rint(on), LO, HI).astype(np.int64)
        nxt = np.roll(on, -1, 0)
        ang_n = np.roll(ang, -1)
        ang_n[-1] += (-2 * np.pi if k % 2 == 0 else 2 * np.pi)
        mid = 0.5 * (ang + ang_n)
        half = 0.5 * np.abs(ang_n - ang)
        rmid = 0.5 * (rad + np.roll(rad, -1)) / np.maximum(np.cos(half), 0.3)
        rmid = rmid * (1 + 0.2 * (uu[2 * Sk:3 * Sk] * 2 - 1))
        ctrl = np.stack([cx0 + rmid * np.cos(mid), cy0 + rmid * np.sin(mid)], 1)
        ctrl = np.clip(np.rint(ctrl), LO, HI).astype(np.int64)
        straight = uu[3 * Sk:4 * Sk] < 0.2
        ctrl[straight] = _div_trunc2(on[straight] + nxt[straight])        # Point.initMiddle
        pts = np.empty((2 * Sk + 1, 2), np.int64)
        pts[0:2 * Sk:2] = on
        pts[1:2 * Sk:2] = ctrl
        pts[2 * Sk] = on[0]
        contours.append(pts.astype(np.int16))
    allp = np.concatenate(contours)
    box = np.array([allp[:, 0].min(), allp[:, 1].min(), allp[:, 0].max(), allp[:, 1].max()], np.int16)
    return contours, box


def synth_glyphset(n_glyphs: int, n_segments: int, first_index: int = 0) -> GlyphSet:
    return _glyphset(synth_glyph, n_glyphs, n_segments, first_index)


def _glyphset(make, n_glyphs: int, n_segments: int, first_index: int = 0) -> GlyphSet:
    pts, cstart, gstart, boxes = [], [0], [0], []
    n = 0
    for g in range(first_index, first_index + n_glyphs):
        cs, box = make(g, n_segments)
        for c in cs:
            pts.append(c)
            n += len(c)
            cstart.append(n)
        gstart.append(len(cstart) - 1)
        boxes.append(box)
    return GlyphSet.from_arrays(np.concatenate(pts), np.array(cstart, np.uint32), np.array(gstart, np.uint32),
                                np.array(boxes, np.int16))


def stroke_glyph(index: int, n_segments: int):
    """Stroke-dense outline (a CJK-like crossing count, not just a CJK-like segment count): 8-16 thin
    elongated closed splines ("strokes": 70 % vertical, 18 % horizontal, 12 % diagonal) scattered over the em,
    all clockwise, so a horizontal ray meets 10-30 edges and strokes that overlap give winding 2 (non-zero fill
    differs from even-odd there).  Same PRNG, contour layout and truncated-midpoint rule as synth_glyph.
    -> (list of (len,2) int16 contour arrays, box[4])"""
    S = int(n_segments)
    assert S >= 32
    rnd = splitmix64(0x57120CE5 ^ index, 8 + 16 * 8 + 4 * S + 64)
    u = _uniform(rnd)
    K = 8 + int(rnd[0] % np.uint64(9))
    while S // K < 4:
        K -= 1
    counts = [S // K] * K
    counts[0] += S - sum(counts)
    contours, o = [], 8
    for k, Sk in enumerate(counts):
        pu = u[o:o + 8]; o += 8
        uu = u[o:o + 4 * Sk]; o += 4 * Sk
        kind = pu[0]
        length = 1000.0 + 800.0 * pu[1]
        thick = 40.0 + 70.0 * pu[2]
        if kind < 0.70:
            theta = np.pi / 2 + (pu[3] * 2 - 1) * 0.06            # vertical stroke
        elif kind < 0.88:
            theta = (pu[3] * 2 - 1) * 0.06                        # horizontal stroke
        else:
            theta = (np.pi / 4 if pu[3] < 0.5 else 3 * np.pi / 4) + (pu[4] * 2 - 1) * 0.2
            length *= 0.8
        cx0 = 250.0 + 1550.0 * pu[5]
        cy0 = 250.0 + 1550.0 * pu[6]
        i = np.arange(Sk)
        ang = -2 * np.pi * (i + uu[0:Sk] * 0.5) / Sk             # clockwise in y-up font units
        ra = 0.5 * length * (1 + 0.06 * (uu[Sk:2 * Sk] * 2 - 1))
        rb = 0.5 * thick * (1 + 0.25 * (uu[Sk:2 * Sk] * 2 - 1))
        ct, st = np.cos(theta), np.sin(theta)

        def place(a, ra_, rb_):
            ex, ey = ra_ * np.cos(a), rb_ * np.sin(a)
            return np.stack([cx0 + ex * ct - ey * st, cy0 + ex * st + ey * ct], 1)
        on = np.clip(np.rint(place(ang, ra, rb)), LO, HI).astype(np.int64)
        nxt = np.roll(on, -1, 0)
        ang_n = np.roll(ang, -1)
        ang_n[-1] -= 2 * np.pi
        mid = 0.5 * (ang + ang_n)
        half = 0.5 * np.abs(ang_n - ang)
        grow = 1.0 / np.maximum(np.cos(half), 0.3) * (1 + 0.1 * (uu[2 * Sk:3 * Sk] * 2 - 1))
        ctrl = np.clip(np.rint(place(mid, 0.5 * (ra + np.roll(ra, -1)) * grow, 0.5 * (rb + np.roll(rb, -1)) * grow)), LO, HI).astype(np.int64)
        straight = uu[3 * Sk:4 * Sk] < 0.2
        ctrl[straight] = _div_trunc2(on[straight] + nxt[straight])        # Point.initMiddle
        pts = np.empty((2 * Sk + 1, 2), np.int64)
        pts[0:2 * Sk:2] = on
        pts[1:2 * Sk:2] = ctrl
        pts[2 * Sk] = on[0]
        contours.append(pts.astype(np.int16))
    allp = np.concatenate(contours)
    box = np.array([allp[:, 0].min(), allp[:, 1].min(), allp[:, 0].max(), allp[:, 1].max()], np.int16)
    return contours, box


def stroke_glyphset(n_glyphs: int, n_segments: int, first_index: int = 0) -> GlyphSet:
    return _glyphset(stroke_glyph, n_glyphs, n_segments, first_index)


def comb_glyph(teeth: int, width: int = 1800, height: int = 1500):
    """A comb with `teeth` vertical teeth: every horizontal ray through the teeth meets
    2*teeth edges — exercises the over-full-row (kmax) fallback.  Straight edges only."""
    x0, y0 = 100, 100
    pitch = width // teeth
    poly = [(x0, y0)]
    for t in range(teeth):
        xa = x0 + t * pitch
        xb = xa + pitch // 2
        poly += [(xa, y0 + height), (xb, y0 + height), (xb, y0 + 200), (xa + pitch, y0 + 200)]
    poly += [(x0 + teeth * pitch, y0)]
    poly = np.array(poly, np.int64)
    nxt = np.roll(poly, -1, 0)
    pts = np.empty((2 * len(poly) + 1, 2), np.int64)
    pts[0:-1:2] = poly
    pts[1:-1:2] = _div_trunc2(poly + nxt)
    pts[-1] = poly[0]
    pts = pts.astype(np.int16)
    box = np.array([pts[:, 0].min(), pts[:, 1].min(), pts[:, 0].max(), pts[:, 1].max()], np.int16)
    return [pts], box
