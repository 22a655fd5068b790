"""CPU suite: the pin of the oracle to the REFERENCE ITSELF — active only once somebody with a Zig >= 0.15 toolchain has
run tools/zig_mint/mint_vectors.zig (inside a checkout of the reference) and committed what it wrote:

    tests/golden/zig_render_glyph.bin   renderGlyph (render_glyph.zig:11-33) of the 190 fixture glyphs, sizes (64, 33, 100, 17)[i % 4]
    tests/golden/zig_lattice.bin        windingInGlyph on Image.GlyphDebug.render's lattice (Image.zig:227-236), 5 glyphs

With the files present, the oracle must equal them byte for byte — that, and nothing else in this repository, turns
"parity unpinned"  into a pinned oracle.  Without them the two comparisons are SKIPPED, loudly.
What always runs: the flat fixture the Zig program reads is the npz fixture (not stale), and the file formats round-trip
through the reader used here (with oracle-made stand-ins written to a temporary directory — never to tests/golden/)."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SIZES = (64, 33, 100, 17)
LATTICE_GLYPHS = (6, 24, 33, 83, 168)          # == LATTICE_GLYPHS of tools/zig_mint/mint_vectors.zig
WHY = ("no Zig-minted vectors in tests/golden/ (the image has no zig toolchain): run tools/zig_mint/mint_vectors.zig where "
       "Zig 0.15 exists — INTEGRATION.md section 6 — and commit its two output files; parity stays UNPINNED until then")


def read_images(path):
    b = open(path, "rb").read()
    assert b[:8] == b"FRZG1\0\0\0", "not a zig_render_glyph.bin"
    (n,), at, out = struct.unpack_from("<I", b, 8), 12, []
    for _ in range(n):
        gi, size, w, h = struct.unpack_from("<IHHH", b, at)
        at += 10
        out.append((gi, size, np.frombuffer(b, np.uint8, w * h, at).reshape(h, w)))
        at += w * h
    assert at == len(b)
    return out


def read_lattices(path):
    b = open(path, "rb").read()
    assert b[:8] == b"FRZL1\0\0\0", "not a zig_lattice.bin"
    (n,), at, out = struct.unpack_from("<I", b, 8), 12, []
    for _ in range(n):
        gi, W, H = struct.unpack_from("<III", b, at)
        at += 12
        out.append((gi, np.frombuffer(b, "<i2", W * H, at).reshape(H, W)))
        at += 2 * W * H
    assert at == len(b)
    return out


def write_images(path, items):
    with open(path, "wb") as f:
        f.write(b"FRZG1\0\0\0" + struct.pack("<I", len(items)))
        for gi, size, im in items:
            f.write(struct.pack("<IHHH", gi, size, im.shape[1], im.shape[0]) + np.ascontiguousarray(im, np.uint8).tobytes())


def write_lattices(path, items):
    with open(path, "wb") as f:
        f.write(b"FRZL1\0\0\0" + struct.pack("<I", len(items)))
        for gi, lat in items:
            f.write(struct.pack("<III", gi, lat.shape[1], lat.shape[0]) + np.ascontiguousarray(lat, "<i2").tobytes())


def test_flat_fixture_is_the_npz_fixture(tmp_path, ascii_set):
    """tests/golden/ascii_glyphs.bin (what the Zig program reads) == a fresh export of ascii_glyphs.npz"""
    have = open(os.path.join(GOLDEN, "ascii_glyphs.bin"), "rb").read()
    assert have[:8] == b"FRFX1\0\0\0" and struct.unpack_from("<I", have, 8)[0] == len(ascii_set) == 190
    at = 12
    for i in range(len(ascii_set)):
        g = ascii_set.glyph(i)
        upm, x0, y0, x1, y1, nc = struct.unpack_from("<H4hI", have, at)
        at += 14
        assert (upm, x0, y0, x1, y1, nc) == (int(ascii_set.g_upm[i]), g.box.x_min, g.box.y_min, g.box.x_max, g.box.y_max, len(g.contours))
        for c in g.contours:
            (npt,) = struct.unpack_from("<I", have, at)
            at += 4
            assert np.array_equal(np.frombuffer(have, "<i2", 2 * npt, at).reshape(npt, 2), c.points)
            at += 4 * npt
    assert at == len(have)
    # the committed script reproduces the committed file
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "zig_mint", "export_fixture.py"), str(tmp_path / "again.bin")],
                          env=env, stdout=subprocess.DEVNULL)
    assert open(tmp_path / "again.bin", "rb").read() == have


def test_vector_formats_round_trip(tmp_path, oracle, ascii_set):
    """the reader above parses files of the documented layout (stand-ins made by the oracle, in a temp dir)"""
    ims = [(i, SIZES[i % 4], oracle.render_glyph(ascii_set.glyph(i), int(ascii_set.g_upm[i]), SIZES[i % 4])) for i in (0, 33, 100)]
    write_images(tmp_path / "a.bin", ims)
    back = read_images(tmp_path / "a.bin")
    assert [(a, b) for a, b, _ in back] == [(a, b) for a, b, _ in ims] and all(np.array_equal(x[2], y[2]) for x, y in zip(back, ims))
    lat = [(83, oracle.winding_lattice(ascii_set.glyph(83)))]
    write_lattices(tmp_path / "l.bin", lat)
    assert np.array_equal(read_lattices(tmp_path / "l.bin")[0][1], lat[0][1])


def test_oracle_equals_zig_render_glyph(oracle, ascii_set):
    path = os.path.join(GOLDEN, "zig_render_glyph.bin")
    if not os.path.exists(path):
        pytest.skip(WHY)
    items = read_images(path)
    assert [g for g, _, _ in items] == list(range(len(ascii_set)))
    for gi, size, zig in items:
        assert size == SIZES[gi % 4]
        mine = oracle.render_glyph(ascii_set.glyph(gi), int(ascii_set.g_upm[gi]), size)
        assert mine.shape == zig.shape, (gi, mine.shape, zig.shape)
        if not np.array_equal(mine, zig):
            y, x = np.argwhere(mine != zig)[0]
            raise AssertionError(f"glyph {gi} ({chr(int(ascii_set.g_char[gi]))!r}) size {size}: pixel ({x}, {y}) oracle {mine[y, x]} != Zig {zig[y, x]}")


def test_oracle_equals_zig_glyph_debug_lattice(oracle, ascii_set):
    path = os.path.join(GOLDEN, "zig_lattice.bin")
    if not os.path.exists(path):
        pytest.skip(WHY)
    items = read_lattices(path)
    assert tuple(g for g, _ in items) == LATTICE_GLYPHS
    for gi, zig in items:
        mine = oracle.winding_lattice(ascii_set.glyph(gi))
        assert mine.shape == zig.shape
        if not np.array_equal(mine, zig):
            y, x = np.argwhere(mine != zig)[0]
            raise AssertionError(f"glyph {gi}: lattice point ({x}, {y}) oracle {mine[y, x]} != Zig {zig[y, x]}")
