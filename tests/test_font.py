"""CPU suite: the C-side contour producer (fr_font_*, font-renderer_amd/csrc/fr_font.cpp) —
SURVEY §8f-1 — against (1) the committed fixture and (2) an independent assembly: raw outlines
from fontTools (a third-party parser, tooling only) pushed through the ORACLE's restatements of
Contour.initTTF (Glyph.zig:43-74) and transform1 (Glyph.zig:178-182)."""
import os

import numpy as np
import pytest

import font_renderer_amd as fr

FONT_DIR = "/usr/local/lib/python3.10/dist-packages/matplotlib/mpl-data/fonts/ttf"


def _font(name, **kw):
    path = os.path.join(FONT_DIR, name)
    if not os.path.exists(path):
        pytest.skip(f"{path} not present")
    return fr.Font.initTTF(path, **kw)


def test_producer_reproduces_fixture(ascii_set):
    for fi, name in enumerate(ascii_set.font_names):
        f = _font(name)
        assert f.information.units_per_em == int(ascii_set.g_upm[np.nonzero(ascii_set.g_font == fi)[0][0]])
        for i in np.nonzero(ascii_set.g_font == fi)[0]:
            g, want = f.getGlyph(int(ascii_set.g_char[i]))[0], ascii_set.glyph(int(i))
            assert g.box == want.box and len(g.contours) == len(want.contours)
            for a, b in zip(g.contours, want.contours):
                assert np.array_equal(a.points, b.points)


def test_hinted_glyph_is_refused_like_the_reference():
    f = _font("DejaVuSans.ttf")                      # 'A' carries instructions: the reference panics (Glyph.zig:85)
    with pytest.raises(fr.FrError) as e:
        f.getGlyph(ord("A"))
    assert e.value.code == -4
    g, _adv = _font("DejaVuSans.ttf", allow_hinted=True).getGlyph(ord("A"))
    assert len(g.contours) == 2 and g.curve_count > 8


def _raw_components(glyf_bytes, off):
    """component records of a composite glyph straight from the file bytes (flags, glyph index,
    raw args as unsigned, raw 2.14 matrix [xscale, scale01, scale10, yscale])"""
    import struct
    pos = off + 10
    parts = []
    while True:
        flags, gidx = struct.unpack(">HH", glyf_bytes[pos:pos + 4]); pos += 4
        if flags & 1:
            a1, a2 = struct.unpack(">HH", glyf_bytes[pos:pos + 4]); pos += 4
        else:
            a1, a2 = glyf_bytes[pos], glyf_bytes[pos + 1]; pos += 2       # zero-extended, as ttf.zig:866 does
        one = 1 << 14
        if flags & 0x0008:
            (sc,) = struct.unpack(">h", glyf_bytes[pos:pos + 2]); pos += 2
            m = [sc, 0, 0, sc]
        elif flags & 0x0040:
            xs, ys = struct.unpack(">hh", glyf_bytes[pos:pos + 4]); pos += 4
            m = [xs, 0, 0, ys]
        elif flags & 0x0080:
            m = list(struct.unpack(">hhhh", glyf_bytes[pos:pos + 8])); pos += 8
        else:
            m = [one, 0, 0, one]
        parts.append((flags, gidx, a1, a2, m))
        if not (flags & 0x0020):
            break
    n_instr = struct.unpack(">H", glyf_bytes[pos:pos + 2])[0] if parts[-1][0] & 0x0100 else 0
    return parts, n_instr


def test_whole_font_simple_and_composite_against_oracle(oracle):
    ft = pytest.importorskip("fontTools.ttLib")
    name = "DejaVuSerif-Italic.ttf"
    f = _font(name)
    tt = ft.TTFont(os.path.join(FONT_DIR, name))
    glyf, order = tt["glyf"], tt.getGlyphOrder()
    raw = tt.reader["glyf"]
    loca = tt["loca"].locations
    cache = {}

    def expected(gi):
        """contours (list of (n,2) int16) per the oracle, or None where the reference would panic"""
        if gi in cache:
            return cache[gi]
        g = glyf[order[gi]]
        res = []
        if g.numberOfContours > 0:
            if len(g.program.getBytecode()) > 0:
                res = None
            else:
                res = oracle.expand_contours(np.array(g.coordinates, np.int16), [fl & 1 for fl in g.flags], list(g.endPtsOfContours))
        elif g.numberOfContours < 0:
            parts, n_instr = _raw_components(raw, loca[gi])
            if n_instr > 0 or any(p[0] & 0x0200 for p in parts):        # Glyph.zig:109,110
                res = None
            for flags, sub_gi, a1, a2, m in ([] if res is None else parts):
                sub = expected(sub_gi)
                if sub is None or not (flags & 0x0002):                  # Glyph.zig:134
                    res = None
                    break
                for c in sub:
                    out = np.zeros_like(c)
                    for k, (x, y) in enumerate(c):
                        rc, ox, oy = oracle.transform_point(int(x), int(y), m, int(np.uint16(a1).astype(np.int16)),
                                                            int(np.uint16(a2).astype(np.int16)), bool(flags & 0x0004))
                        if rc != 0:
                            res = None
                            break
                        out[k] = (ox, oy)
                    if res is None:
                        break
                    res.append(out)
                if res is None:
                    break
        cache[gi] = res
        return res

    n_simple = n_comp = n_refused = 0
    for gi, gname in enumerate(order):
        want = expected(gi)
        if want is None:
            with pytest.raises(fr.FrError):
                f.glyph_by_index(gi)
            n_refused += 1
            continue
        got = f.glyph_by_index(gi)
        assert len(got.contours) == len(want), gname
        for a, b in zip(got.contours, want):
            assert np.array_equal(a.points, b), gname
        if glyf[gname].numberOfContours < 0:
            n_comp += 1
        elif glyf[gname].numberOfContours > 0:
            n_simple += 1
            g = glyf[gname]
            assert got.box == fr.Box(g.xMin, g.yMin, g.xMax, g.yMax)
    assert n_simple > 1500 and n_comp + n_refused > 1000, (n_simple, n_comp, n_refused)
    print(f"simple {n_simple}, composite {n_comp}, refused like the reference {n_refused}")


def test_cmap_lookup_matches_fonttools():
    ft = pytest.importorskip("fontTools.ttLib")
    for name in ("STIXGeneral.ttf", "DejaVuSerif-Italic.ttf"):
        f = _font(name)
        tt = ft.TTFont(os.path.join(FONT_DIR, name))
        cmap, order = tt.getBestCmap(), tt.getGlyphOrder()
        gid = {n: i for i, n in enumerate(order)}
        for ch in list(range(0x20, 0x250)) + [0x3b1, 0x2202, 0x221e, 0xfb01, 0x1d49c]:
            if ch > 0xFFFF:
                continue                     # the chosen subtable may be BMP-only (format 4)
            want = gid[cmap[ch]] if ch in cmap else 0
            assert f.glyph_index(ch) == want, (name, hex(ch))


def test_qoi_writer_matches_oracle_and_decodes(oracle, ascii_set):
    """fr_qoi_* (product, C++) == oracle restatement of qoi.zig byte for byte; gray path == RGB path"""
    from font_renderer_amd.qoi import saveRGB
    from test_oracle import _qoi_decode
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (33, 41, 3), dtype=np.uint8)
    img[5:20] = img[5, 0]
    img[21, :, :] = (np.arange(41)[:, None] * 2) % 256
    assert saveRGB(img) == oracle.qoi_encode(img)
    assert np.array_equal(_qoi_decode(saveRGB(img)), img)
    g = ascii_set.glyph(ascii_set.find("STIX", "R"))
    gray = oracle.render_glyph(g, 1000, 80)
    rgb = np.repeat(gray[:, :, None], 3, 2)
    assert saveRGB(gray) == oracle.qoi_encode(rgb) == saveRGB(rgb)
    dbg = oracle.glyph_debug_render(g, 50)
    assert saveRGB(dbg) == oracle.qoi_encode(dbg)


def test_atlas_pages():
    from font_renderer_amd.atlas import atlas_pages
    assert atlas_pages(95, 128) == [(0, 95)]
    assert atlas_pages(600, 128) == [(0, 256), (256, 256), (512, 88)]


@pytest.mark.parametrize("name", ["DejaVuSerif-Italic.ttf", "STIXGeneral.ttf", "DejaVuSansMono.ttf"])
def test_advance_widths_follow_the_reference_reading_of_hmtx(name):
    """Font.getGlyph's second result (Font.zig:161-169): loadAdvanceWidths (:123-139) reads the advance of each of the
    first numberOfHMetrics glyphs (as i16) and, for the glyphs after them, the i16 entries that FOLLOW the long metrics —
    which are left-side bearings in the TrueType layout.  fr_font_glyph_advance reproduces exactly that; checked against
    fontTools' independent parse of hhea / hmtx."""
    ft = pytest.importorskip("fontTools.ttLib")
    path = os.path.join(FONT_DIR, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not in this image")
    f = _font(name, allow_hinted=True)
    tt = ft.TTFont(path)
    order, hmtx, n_long = tt.getGlyphOrder(), tt["hmtx"], tt["hhea"].numberOfHMetrics
    assert f.num_glyphs == len(order)
    for gi, gname in enumerate(order):
        adv, lsb = hmtx[gname]
        want = adv if gi < n_long else lsb
        want = want - 65536 if want > 32767 else want                     # (:129 reads the u16 advance as i16)
        assert f.advance_width(gi) == want, (gi, gname, n_long)
    gi = f.glyph_index(ord("A"))
    g, adv = f.getGlyph(ord("A"))
    # (a monospaced font keeps few long metrics: DejaVuSansMono's 'A' is glyph 36 of numberOfHMetrics 2, and gets its
    # left-side bearing 37 as "advance" where the font says 1233 — the reference's reading, reproduced)
    assert adv == f.advance_width(gi) == (hmtx[order[gi]][0] if gi < n_long else hmtx[order[gi]][1]) and len(g.contours) >= 1
    with pytest.raises(Exception):
        f.advance_width(f.num_glyphs)
