"""CPU suite: pins the oracle (oracle/fr_oracle.c) as far as it can be pinned.

The reference ships no golden vectors, so these tests check the C
restatement against (1) the committed fixture's independently computed contour
expansion, (2) SURVEY Appendix B's known answers (an independent numpy emulation made
at survey time), (3) tests/ref_numpy.py, a second restatement written separately, and
(4) internal consistency the reference itself exhibits (f32 path == integer path on
lattice points).  PARITY UNPINNED against the Zig binary — stated in DESIGN.md."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import ref_numpy
from font_renderer_amd.glyph import Box, Contour, Glyph
from font_renderer_amd.synth import comb_glyph, synth_glyph

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_contour_expansion_matches_fixture(oracle, ascii_set):
    """or_contour_init_ttf == the fixture script's independent Contour.initTTF (Glyph.zig:43-74)"""
    for i in range(len(ascii_set)):
        coords, on, ends = ascii_set.raw(i)
        g = ascii_set.glyph(i)
        if len(ends) == 0:
            assert g.contours == []
            continue
        got = oracle.expand_contours(coords, on, ends)
        assert len(got) == len(g.contours)
        for a, b in zip(got, g.contours):
            assert np.array_equal(a, b.points)
            assert len(a) % 2 == 1 and np.array_equal(a[0], a[-1])       # Glyph.zig:23


def test_contour_expansion_hand_cases(oracle):
    # all on-curve square: every edge gets a truncated midpoint (geometry.zig:12-17)
    sq = oracle.expand_contours([(0, 0), (0, 11), (11, 11), (11, 0)], [1, 1, 1, 1], [3])[0]
    assert sq.tolist() == [[11, 0], [5, 0], [0, 0], [0, 5], [0, 11], [5, 11], [11, 11], [11, 5], [11, 0]]
    # negative coordinates truncate toward zero, not floor: (-3 + 0)/2 = -1
    tr = oracle.expand_contours([(-3, -3), (0, 4), (4, -3)], [1, 1, 1], [2])[0]
    assert tr[1].tolist() == [0, -3] and tr[3].tolist() == [-1, 0]
    # all off-curve: starts at index 0 with an implied point, closes by appending buf[0] (Glyph.zig:69-71)
    off = oracle.expand_contours([(0, 0), (10, 0), (10, 10), (0, 10)], [0, 0, 0, 0], [3])[0]
    assert len(off) == 9 and np.array_equal(off[0], off[-1]) and off[0].tolist() == [0, 5]


def test_transform1(oracle):
    ident = [1 << 14, 0, 0, 1 << 14]
    # Glyph.zig:180: |a|-|c| > 8 -> shift not doubled; tmp*shift = 16384*e
    assert oracle.transform_point(5, 7, ident, 100, -50, True) == (0, 105, -43)
    # |a| == |c| within 8 -> e doubled (the reference's rule as written)
    m = [1 << 13, 0, 1 << 13, 1 << 14]
    rc, x, _ = oracle.transform_point(4, 4, m, 10, 0, True)
    assert rc == 0 and x == (((1 << 13) * 4 + (1 << 13) * 4 + (1 << 13) * 20 + (1 << 13)) >> 14)
    # not round_to_grid and a fractional result -> the reference @panic("not impl")
    assert oracle.transform_point(1, 0, [1 << 13, 0, 0, 1 << 14], 0, 0, False)[0] == -1


def test_appendix_b_stix_A(oracle, ascii_set):
    """SURVEY Appendix B known answers for STIXGeneral 'A' at font_size 64"""
    i = ascii_set.find("STIX", "A")
    g = ascii_set.glyph(i)
    assert ascii_set.gs.boxes[i].tolist() == [15, 0, 707, 674] and int(ascii_set.g_upm[i]) == 1000
    assert len(g.contours) == 2 and g.curve_count == 25
    _, _, w, h, _ = oracle.render_glyph_dims(g.box.as_array(), 1000, 64)
    assert (w, h) == (47, 45)
    gray = oracle.render_glyph(g, 1000, 64)
    mn, mx, _, _, scale = oracle.render_glyph_dims(g.box.as_array(), 1000, 64)
    wd = oracle.render_cell(g, mn[0], mx[1], w, h, scale, O.WINDING_I16)
    hist = {int(v): int(c) for v, c in zip(*np.unique(wd, return_counts=True))}
    assert hist == {-2: 1, -1: 28, 0: 1641, 1: 445}
    assert set(np.nonzero(wd < 0)[0].tolist()) == {h - 1}           # all negatives on the y = 0 row (F6)
    assert hashlib.sha256(gray.tobytes()).hexdigest().startswith("0417227156fe3a7d")
    assert hashlib.sha256(wd.astype("<i2").tobytes()).hexdigest().startswith("aaf7da9eb8cd1a72")
    assert np.array_equal(gray, np.clip(wd.astype(int) * 20 + 100, 0, 255).astype(np.uint8))
    # DejaVuSerif-Italic 'A'
    j = ascii_set.find("DejaVu", "A")
    assert ascii_set.gs.boxes[j].tolist() == [-158, 0, 1375, 1493] and ascii_set.glyph(j).curve_count == 19


def test_appendix_b_integer_rows(oracle, ascii_set):
    g = ascii_set.glyph(ascii_set.find("STIX", "A"))
    xs = np.arange(14, 709)
    ct, _ = oracle.glyph_info(g)
    names = ["x_axis", "balance", "up_stright", "up_normal", "up_u", "up_inv_u", "down_stright", "down_normal", "down_inv_u", "down_u"]
    cls = {names[k]: int((ct == k).sum()) for k in range(10) if (ct == k).any()}
    assert cls == {"down_stright": 5, "x_axis": 5, "up_stright": 5, "up_normal": 5, "down_normal": 5}
    assert int(oracle.glyph_info(g)[1].sum()) == 19
    for y, want in [(0, {-2: 2, -1: 437, 0: 256}), (674, {1: 354, 0: 341})]:
        wi = oracle.winding_in_glyph(g, np.stack([xs, np.full_like(xs, y)], 1))
        wf = np.array([oracle.winding_at(g, float(x), float(y)) for x in xs], np.int16)
        assert np.array_equal(wi, wf)
        assert {int(v): int(c) for v, c in zip(*np.unique(wi, return_counts=True))} == want
    for y in (1, 16, 673):
        wi = oracle.winding_in_glyph(g, np.stack([xs, np.full_like(xs, y)], 1))
        assert set(np.unique(wi).tolist()) <= {0, 1}
    assert oracle.diag() == (0, 0)      # inside the reference's i64 domain, assert(dy>=0) never violated


def test_second_restatement_agrees(oracle, ascii_set):
    """oracle/fr_oracle.c == tests/ref_numpy.py on whole images, several glyphs and sizes"""
    for font, ch, size in [("STIX", "A", 64), ("STIX", "g", 40), ("DejaVu", "A", 64), ("DejaVu", "Q", 33),
                           ("STIX", "%", 50), ("DejaVu", "@", 71), ("STIX", " ", 20)]:
        i = ascii_set.find(font, ch)
        g, upm = ascii_set.glyph(i), int(ascii_set.g_upm[i])
        gs1 = ascii_set.gs.subset(i, i + 1)
        wd, gray = ref_numpy.render_glyph(gs1.points_xy, gs1.contour_start, gs1.boxes[0], upm, size)
        assert np.array_equal(gray, oracle.render_glyph(g, upm, size))
    cs, box = synth_glyph(7, 64)
    g = Glyph(Box(*[int(v) for v in box]), [Contour(c) for c in cs])
    from font_renderer_amd.glyph import GlyphSet
    gs1 = GlyphSet([g])
    _, gray = ref_numpy.render_glyph(gs1.points_xy, gs1.contour_start, gs1.boxes[0], 2048, 96)
    assert np.array_equal(gray, oracle.render_glyph(g, 2048, 96))


def test_f32_path_equals_integer_path_on_lattice(oracle, ascii_set):
    """Appendix B: on the step-14 sub-lattice of the GlyphDebug grid of STIX 'A' (2 450 points,
    x = x_min-1+14i, y = y_max+1-14j) the f32 path and the integer path agree everywhere.
    (Not a general law: e.g. (420, 321) sits on an almost-straight edge that the integer path
    treats as a line (.up_stright, render_glyph.zig:99) and the f32 path as a parabola.)"""
    oracle.diag_reset()
    g = ascii_set.glyph(ascii_set.find("STIX", "A"))
    b = g.box
    xs, ys = np.arange(b.x_min - 1, b.x_max + 2, 14), np.arange(b.y_max + 1, b.y_min - 2, -14)
    q = np.stack(np.meshgrid(xs, ys), -1).reshape(-1, 2)
    assert len(q) == 2450
    wi = oracle.winding_in_glyph(g, q)
    wf = np.array([oracle.winding_at(g, float(x), float(y)) for x, y in q], np.int16)
    assert np.array_equal(wi, wf)
    assert oracle.winding_in_glyph(g, [(420, 321)])[0] == 0 and oracle.winding_at(g, 420.0, 321.0) == 1
    assert oracle.diag() == (0, 0)


def test_coverage_definition(oracle, ascii_set):
    g = ascii_set.glyph(ascii_set.find("STIX", "B"))
    mn, mx, w, h, scale = oracle.render_glyph_dims(g.box.as_array(), 1000, 48)
    wd = oracle.render_cell(g, mn[0], mx[1], w, h, scale, O.WINDING_I16)
    mask = oracle.render_cell(g, mn[0], mx[1], w, h, scale, O.MASK_NONZERO)
    cov1 = oracle.render_cell(g, mn[0], mx[1], w, h, scale, O.COVERAGE_U8, 1)
    assert np.array_equal(mask, np.where(wd != 0, 255, 0).astype(np.uint8)) and np.array_equal(cov1, mask)
    # n = 2 corner phase: sub-sample (0,0) is the reference sample; coverage is the mean of 4 masks
    cov2 = oracle.render_cell(g, mn[0], mx[1], w, h, scale, O.COVERAGE_U8, 2)
    assert set(np.unique(cov2).tolist()) <= {0, 64, 128, 191, 255}
    cov4 = oracle.render_cell(g, mn[0], mx[1], w, h, scale, O.COVERAGE_U8, 4, True)
    assert cov4.max() == 255 and cov4.min() == 0 and 0 < (cov4 > 0).mean() < 1


def test_empty_glyph_is_1x1_of_100(oracle):
    g = Glyph.initEmpty()
    out = oracle.render_glyph(g, 1000, 64)
    assert out.shape == (1, 1) and out[0, 0] == 100


def test_image_colour_maps(oracle):
    assert oracle.winding_rgb(0, 50, 150) == (0, 0, 0)
    assert oracle.winding_rgb(1, 50, 150) == (0, 0, 50) and oracle.winding_rgb(-2, 50, 150) == (100, 0, 0)
    assert oracle.winding_rgb(6, 50, 150) == (150, 150, 255)       # 300 > 255: overflow tint
    from font_renderer_amd.image import Winding
    im = Winding.init(4, 1, 50, 150)
    im.data[:] = [0, 1, -2, 6]
    assert [im.getRGBLinear(i) for i in range(4)] == [oracle.winding_rgb(int(v), 50, 150) for v in im.data]


def _qoi_decode(buf: bytes):
    assert buf[:4] == b"qoif" and buf[12] == 3 and buf[13] == 0 and buf[-8:] == bytes(7) + b"\x01"
    w, h = int.from_bytes(buf[4:8], "big"), int.from_bytes(buf[8:12], "big")
    px, run_tab, out, i = (0, 0, 0), [(0, 0, 0)] * 64, [], 14
    while len(out) < w * h:
        b = buf[i]; i += 1
        if b == 0xFE:
            px = tuple(buf[i:i + 3]); i += 3
        elif b >> 6 == 0:
            px = run_tab[b]
        elif b >> 6 == 1:
            px = tuple((p + d - 2) & 255 for p, d in zip(px, ((b >> 4) & 3, (b >> 2) & 3, b & 3)))
        elif b >> 6 == 2:
            b2 = buf[i]; i += 1
            vg = (b & 63) - 32
            px = ((px[0] + vg - 8 + (b2 >> 4)) & 255, (px[1] + vg) & 255, (px[2] + vg - 8 + (b2 & 15)) & 255)
        else:
            out += [px] * (b & 63)
        out.append(px)
        run_tab[(px[0] * 3 + px[1] * 5 + px[2] * 7 + 255 * 11) % 64] = px
    return np.array(out[:w * h], np.uint8).reshape(h, w, 3)


def test_qoi_writer_roundtrip(oracle, ascii_set):
    """qoi.zig:25-88 restatement decodes back (standard QOI decoder, RGB, alpha 255)"""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (17, 23, 3), dtype=np.uint8)
    img[3:9] = img[3, 0]                      # long runs (> 62) and index hits
    img[10, :, :] = np.arange(23)[:, None]    # small diffs
    assert np.array_equal(_qoi_decode(oracle.qoi_encode(img)), img)
    g = ascii_set.glyph(ascii_set.find("STIX", "i"))
    dbg = oracle.glyph_debug_render(g, 50)
    assert np.array_equal(_qoi_decode(oracle.qoi_encode(dbg)), dbg)
    assert (dbg == (255, 255, 0)).all(-1).sum() > 0 and (dbg == (0, 255, 255)).all(-1).sum() > 0


def test_known_answers(oracle, ascii_set):
    """self-minted known answers (tests/golden/known_answers.json; regenerate with
    tests/golden/make_known_answers.py): guards the oracle against silent edits."""
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        ka = json.load(f)
    import sys
    sys.path.insert(0, GOLDEN)
    import make_known_answers
    got = make_known_answers.compute(oracle, ascii_set)
    assert got == ka


def test_exact_lattice_k_twin(oracle, ascii_set):
    """SURVEY §8 f-3 (build-defined, no reference semantics): K = 1 at the GlyphDebug origin is
    or_winding_lattice; a glyph whose straight edges carry EXACT midpoints has the same winding at
    K * p on the K-scaled glyph as at p on the original (only the +-1 'straight' tolerance of
    render_glyph.zig:99,103 depends on the scale)"""
    g = ascii_set.glyph(ascii_set.find("STIX", "A"))
    box = g.box.as_array().astype(int)
    W, H = box[2] - box[0] + 3, box[3] - box[1] + 3
    assert np.array_equal(oracle.exact_lattice(g, 1, box[0] - 1, box[3] + 1, W, H), oracle.winding_lattice(g))
    from font_renderer_amd.glyph import Box, Contour, Glyph
    sq = np.array([(0, 0), (0, 10), (0, 20), (10, 20), (20, 20), (20, 10), (20, 0), (10, 0), (0, 0)], np.int16)
    gq = Glyph(Box(0, 0, 20, 20), [Contour(sq)])
    l1 = oracle.exact_lattice(gq, 1, -2, 22, 25, 25)
    l4 = oracle.exact_lattice(gq, 4, -8, 88, 97, 97)
    assert np.array_equal(l4[::4, ::4], l1)
    cov = oracle.exact_coverage(gq, 4, -8, 88, 24, 24, 4)
    assert cov[10, 10] == 255 and cov[0, 0] == 0
