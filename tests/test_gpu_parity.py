"""GPU suite: the HIP path (through the C ABI) against the oracle, bit for bit.

Bars (BASELINE.json north_star): integer winding counts bit-exact; 8-bit coverage within
1 LSB — these tests demand 0 LSB because every sub-sample's inside test is exact."""
import numpy as np
import pytest

import font_renderer_amd as fr
import oracle_lib as O
from font_renderer_amd import render_glyph as rg
from font_renderer_amd.atlas import atlas_shape, cell_jobs
from font_renderer_amd.glyph import Box, Contour, Glyph, GlyphSet
from font_renderer_amd.synth import comb_glyph, stroke_glyphset, synth_glyphset

pytestmark = pytest.mark.gpu

MODES = [(fr.FR_WINDING_I16, O.WINDING_I16), (fr.FR_GRAY_DEBUG, O.GRAY_DEBUG), (fr.FR_MASK_NONZERO, O.MASK_NONZERO)]


def _batch_both(ctx, oracle, gs, jobs, mode, shape, n=1, center=False, threads=8, dgs=None):
    own = dgs is None
    dgs = dgs or fr.DeviceGlyphSet(ctx, gs)
    dt = np.int16 if mode == fr.FR_WINDING_I16 else np.uint8
    got = np.full(shape, 7, dt)
    ref = np.full(shape, 7, dt)
    rg.render_batch(dgs, jobs, mode, got, n, fr.FR_SAMPLE_CENTER if center else fr.FR_SAMPLE_CORNER)
    oracle.render_batch(gs, jobs, mode, ref, n, center, threads)
    if own:
        dgs.close()
    return got, ref


def test_render_glyph_all_ascii_fixture(ctx, oracle, ascii_set):
    """renderGlyph drop-in (render_glyph.zig:11-33) on all 190 real glyphs, incl. the
    degenerate rows the reference mis-counts and the empty glyph (' ')."""
    bad = []
    for i in range(len(ascii_set)):
        g, upm = ascii_set.glyph(i), int(ascii_set.g_upm[i])
        size = (64, 33, 100, 17)[i % 4]
        im = fr.renderGlyph(g, fr.FontInformation(upm), size, ctx=ctx)
        want = oracle.render_glyph(g, upm, size)
        if (im.height, im.width) != want.shape or not np.array_equal(im.as_2d(), want):
            bad.append((i, chr(int(ascii_set.g_char[i])), size))
    assert not bad, bad


def test_render_glyph_stix_A_known_answer(ctx, ascii_set):
    """SURVEY Appendix B straight from the GPU: 47x45, histogram, negatives on the y=0 row"""
    i = ascii_set.find("STIX", "A")
    wd = rg.renderGlyphWinding(ascii_set.glyph(i), fr.FontInformation(1000), 64, ctx=ctx)
    assert (wd.width, wd.height) == (47, 45)
    hist = {int(v): int(c) for v, c in zip(*np.unique(wd.data, return_counts=True))}
    assert hist == {-2: 1, -1: 28, 0: 1641, 1: 445}
    assert set(np.nonzero(wd.as_2d() < 0)[0].tolist()) == {44}


@pytest.mark.parametrize("mode,omode", MODES)
def test_modes_atlas_ascii(ctx, oracle, ascii_set, mode, omode):
    """configs[1] shape: 95 glyphs, 128x128 cells, one 2048^2 atlas (n = 1 modes)"""
    gs = ascii_set.gs
    jobs = cell_jobs(gs, 128, 100, ascii_set.g_upm, 16, first_glyph=0, n_glyphs=95)
    got, ref = _batch_both(ctx, oracle, gs, jobs, mode, (2048, 2048))
    assert np.array_equal(got, ref)
    assert (got[6 * 128:, :] == 7).all()        # cells beyond the 95th are not touched


@pytest.mark.parametrize("n,center", [(1, False), (1, True), (2, False), (2, True), (4, False), (4, True)])
def test_coverage_atlas_ascii(ctx, oracle, ascii_set, n, center):
    gs = ascii_set.gs
    jobs = cell_jobs(gs, 64, 50, ascii_set.g_upm, 16, first_glyph=95, n_glyphs=95)   # DejaVuSerif-Italic
    got, ref = _batch_both(ctx, oracle, gs, jobs, fr.FR_COVERAGE_U8, (6 * 64, 16 * 64), n, center)
    diff = np.abs(got.astype(int) - ref.astype(int))
    assert diff.max() == 0, f"max |coverage error| = {diff.max()} LSB at {np.argwhere(diff > 0)[:5].tolist()}"


@pytest.mark.parametrize("segs,cell", [(16, 64), (32, 128), (64, 96), (128, 256), (256, 128)])
def test_coverage_synthetic(ctx, oracle, segs, cell):
    """SURVEY §8d synthetic outlines, 4x4 centre samples (configs[2]/[3] shapes, reduced count)"""
    n_g = 8 if cell < 200 else 4
    gs = synth_glyphset(n_g, segs, first_index=100 + segs)
    jobs = cell_jobs(gs, cell, cell, 2048, 4)
    got, ref = _batch_both(ctx, oracle, gs, jobs, fr.FR_COVERAGE_U8, atlas_shape(n_g, cell, 4), 4, True)
    assert np.array_equal(got, ref)
    assert 0.05 < (ref > 0).mean() < 0.95


def test_mixed_glyph_sizes_in_one_plan(ctx, oracle):
    """The kernel decision is per JOB: one plan holding glyphs of 24, 128, 129, 200, 520 and 800 segments (520: the
    1024-record instances of the fast kernels; 800: the general kernel, four 256-record chunks) renders every cell like the
    oracle; the small glyphs keep taking the in-kernel record build whatever else is in the set."""
    parts = [synth_glyphset(2, 24, first_index=11), synth_glyphset(2, 128, first_index=12),
             synth_glyphset(1, 129, first_index=13), synth_glyphset(2, 200, first_index=14),
             synth_glyphset(1, 520, first_index=15), synth_glyphset(2, 64, first_index=16), synth_glyphset(1, 800, first_index=17)]
    gl = [p.glyph(i) for p in parts for i in range(len(p))]
    gs = GlyphSet(gl)
    for cell, n in ((128, 4), (256, 4), (96, 2), (80, 1)):
        jobs = cell_jobs(gs, cell, cell, 2048, 5)
        got, ref = _batch_both(ctx, oracle, gs, jobs, fr.FR_COVERAGE_U8, atlas_shape(len(gs), cell, 5), n, True, threads=16)
        assert np.array_equal(got, ref), (cell, n)
        if cell in (128, 256):
            # this plan holds all five kinds of job (<= 128 / 256 / 512 / 1024 candidate roots: cov4_kernel with 2 / 4 / 8 / 16
            # records per lane; 800 segments: the general kernel, forked onto the context's second stream) — the same bytes in one stream
            st = fr.Plan(fr.DeviceGlyphSet(ctx, gs), jobs, fr.FR_COVERAGE_U8, n, fr.FR_SAMPLE_CENTER).stats()
            assert st["jobs_cov4"] == len(gs) - 1 and st["jobs_general"] == 1, st
            assert np.array_equal(_render_opt(ctx, gs, jobs, atlas_shape(len(gs), cell, 5), n, True, overlap=0), ref)
            assert np.array_equal(_render_opt(ctx, gs, jobs, atlas_shape(len(gs), cell, 5), n, True, overlap=2), ref)   # (forked whatever the size)
    jobs = cell_jobs(gs, 64, 64, 2048, 5)
    got, ref = _batch_both(ctx, oracle, gs, jobs, fr.FR_WINDING_I16, atlas_shape(len(gs), 64, 5), 1, False, threads=16)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("segs,cell", [(64, 128), (128, 256), (200, 256)])
def test_coverage_stroke_dense(ctx, oracle, segs, cell):
    """stroke-dense outlines (8-16 thin strokes per glyph: 10-30 crossings per ray, overlapping strokes
    give winding 2), 16 samples per pixel: every sort tier and over-full rows at the default kmax"""
    n_g = 6
    gs = stroke_glyphset(n_g, segs, first_index=300 + segs)
    jobs = cell_jobs(gs, cell, cell, 2048, 3)
    got, ref = _batch_both(ctx, oracle, gs, jobs, fr.FR_COVERAGE_U8, atlas_shape(n_g, cell, 3), 4, True, threads=16)
    assert np.array_equal(got, ref)
    w, wref = _batch_both(ctx, oracle, gs, jobs, fr.FR_WINDING_I16, atlas_shape(n_g, cell, 3), 1, False, threads=16)
    assert np.array_equal(w, wref)
    assert wref.max() >= 2          # overlapping strokes: non-zero fill is not even-odd here


def _render_opt(ctx, gs, jobs, shape, n, center, **opts):
    """one coverage render with context options set for its duration"""
    defaults = {"cov4": 1, "kmax": 32, "min_wgs": 2048, "strip_px": 256, "overlap": 1}
    try:
        for k, v in opts.items():
            ctx.set_option(k, v)
        dgs = fr.DeviceGlyphSet(ctx, gs)
        got = np.full(shape, 7, np.uint8)
        rg.render_batch(dgs, jobs, fr.FR_COVERAGE_U8, got, n, fr.FR_SAMPLE_CENTER if center else fr.FR_SAMPLE_CORNER)
        dgs.close()
    finally:
        for k in opts:
            ctx.set_option(k, defaults[k])
    return got


@pytest.mark.parametrize("cell,segs,center", [(256, 128, True), (256, 128, False), (128, 32, True), (128, 200, False), (256, 250, True)])
def test_cov4_kernel_matches_the_oracle(ctx, oracle, cell, segs, center):
    """cov4_kernel (fr_cov4.hip: coverage by integrating signed byte differences, 4-wave workgroups) on the
    shapes it takes — 256- and 128-pixel cells, 16 samples per pixel, up to 256 segments — against the
    oracle, and byte-identical to the general kernel on the same plan (ctx option cov4 = 0)."""
    n_g = 6
    gs = synth_glyphset(n_g, segs, first_index=500 + segs + cell)
    jobs = cell_jobs(gs, cell, cell, 2048, 3)
    shape = atlas_shape(n_g, cell, 3)
    got = _render_opt(ctx, gs, jobs, shape, 4, center)
    ref = np.full(shape, 7, np.uint8)
    oracle.render_batch(gs, jobs, O.COVERAGE_U8, ref, 4, center, 16)
    assert np.array_equal(got, ref)
    assert np.array_equal(_render_opt(ctx, gs, jobs, shape, 4, center, cov4=0), ref)
    # bands split over workgroups (few jobs would not fill the chip) and kmax tiers (over-full rows by direct sum)
    assert np.array_equal(_render_opt(ctx, gs, jobs, shape, 4, center, min_wgs=1 << 20), ref)
    assert np.array_equal(_render_opt(ctx, gs, jobs, shape, 4, center, kmax=8), ref)
    assert np.array_equal(_render_opt(ctx, gs, jobs, shape, 4, center, kmax=16, min_wgs=1), ref)


def test_cov4_kernel_strips_combs_strokes_and_mixed_plans(ctx, oracle):
    """cov4_kernel: cells wider than one strip (512 x 64 px as two 256-px strips, 256 x 32 as two 128-px
    strips), over-full rows (combs: 80 crossings per ray), stroke-dense glyphs, and a plan that mixes cells
    it takes with cells it does not (odd sizes, > 256 segments): the job table is reordered inside the plan,
    every cell still lands where its job says."""
    # two strips per cell
    gs = synth_glyphset(3, 64, first_index=880)
    rows = [(i, int(np.floor(gs.boxes[i][0] * 0.25)), int(np.ceil(gs.boxes[i][3] * 0.25)) - 200, 512, 64, 0, 64 * i, np.float32(0.25)) for i in range(3)]
    jobs = rg.make_jobs(rows)
    ref = np.full((192, 512), 7, np.uint8)
    oracle.render_batch(gs, jobs, O.COVERAGE_U8, ref, 4, True, 16)
    assert np.array_equal(_render_opt(ctx, gs, jobs, (192, 512), 4, True), ref)
    assert np.array_equal(_render_opt(ctx, gs, jobs, (192, 512), 4, True, strip_px=128), ref)
    assert ref.max() == 255
    # combs and strokes
    gl = []
    for teeth in (40, 6):
        cs, box = comb_glyph(teeth)
        gl.append(Glyph(Box(*[int(v) for v in box]), [Contour(c) for c in cs]))
    sg = stroke_glyphset(4, 160, first_index=41)
    gl += [sg.glyph(i) for i in range(len(sg))]
    gs = GlyphSet(gl)
    for cell in (128, 256):
        jobs = cell_jobs(gs, cell, cell - 16, 2048, 3)
        shape = atlas_shape(len(gs), cell, 3)
        ref = np.full(shape, 7, np.uint8)
        oracle.render_batch(gs, jobs, O.COVERAGE_U8, ref, 4, True, 16)
        for kmax in (32, 16, 8):
            assert np.array_equal(_render_opt(ctx, gs, jobs, shape, 4, True, kmax=kmax), ref), (cell, kmax)
    # mixed plan
    parts = [synth_glyphset(3, 100, first_index=21), synth_glyphset(2, 300, first_index=22), synth_glyphset(3, 256, first_index=23)]
    gs = GlyphSet([p.glyph(i) for p in parts for i in range(len(p))])
    jobs = cell_jobs(gs, 128, 128, 2048, 4).copy()
    jobs["w"][1] = 125                      # not a whole strip: general kernel
    jobs["h"][5] = 120                      # not a whole wave band
    shape = atlas_shape(len(gs), 128, 4)
    ref = np.full(shape, 7, np.uint8)
    oracle.render_batch(gs, jobs, O.COVERAGE_U8, ref, 4, True, 16)
    assert np.array_equal(_render_opt(ctx, gs, jobs, shape, 4, True), ref)


@pytest.mark.parametrize("mode,omode", MODES)
def test_win1_kernel_matches_the_oracle(ctx, oracle, mode, omode):
    """win1_kernel (fr_win1.hip: one sample per pixel, crossings added straight into a row of winding
    differences, no sort): Image.Winding values, renderGlyph's gray map and the non-zero mask on 256- and
    128-pixel cells — synthetic, stroke-dense (windings up to 4+), combs (rows of 80 crossings: the direct
    path), two strips per cell, bands split over workgroups — against the oracle and the general kernel"""
    gl = []
    for teeth in (40, 6):
        cs, box = comb_glyph(teeth)
        gl.append(Glyph(Box(*[int(v) for v in box]), [Contour(c) for c in cs]))
    parts = [synth_glyphset(3, 128, first_index=61), stroke_glyphset(3, 160, first_index=62), synth_glyphset(2, 250, first_index=63)]
    gl += [p.glyph(i) for p in parts for i in range(len(p))]
    # ten nested squares of one orientation, both ways round: windings up to +10 and down to -10 — both clamps of the
    # gray map (render_glyph.zig:28: w >= 8 -> 255, w <= -5 -> 0), i.e. the map's general path
    for flip in (False, True):
        cs = []
        for k in range(10):
            a, b = 100 + 90 * k, 1900 - 90 * k
            poly = np.array([(a, a), (a, b), (b, b), (b, a)][::-1 if flip else 1], np.int64)
            nxt = np.roll(poly, -1, 0)
            pts = np.empty((9, 2), np.int64)
            pts[0:-1:2] = poly
            pts[1:-1:2] = np.trunc((poly + nxt) / 2)
            pts[-1] = poly[0]
            cs.append(Contour(pts.astype(np.int16)))
        gl.append(Glyph(Box(100, 100, 1900, 1900), cs))
    gs = GlyphSet(gl)
    dt = np.int16 if mode == fr.FR_WINDING_I16 else np.uint8
    for cell, center in ((256, False), (128, True)):
        jobs = cell_jobs(gs, cell, cell - 20, 2048, 4)
        shape = atlas_shape(len(gs), cell, 4)
        ref = np.full(shape, 7, dt)
        oracle.render_batch(gs, jobs, omode, ref, 1, center, 16)
        for opts in ({}, {"cov4": 0}, {"min_wgs": 1 << 20}):
            try:
                for k, v in opts.items():
                    ctx.set_option(k, v)
                dgs = fr.DeviceGlyphSet(ctx, gs)
                got = np.full(shape, 7, dt)
                plan = fr.Plan(dgs, jobs, mode, 1, fr.FR_SAMPLE_CENTER if center else fr.FR_SAMPLE_CORNER)
                assert plan.stats()["jobs_cov4"] == (0 if "cov4" in opts else len(gs))
                plan.close()
                rg.render_batch(dgs, jobs, mode, got, 1, fr.FR_SAMPLE_CENTER if center else fr.FR_SAMPLE_CORNER)
                dgs.close()
            finally:
                ctx.set_option("cov4", 1); ctx.set_option("min_wgs", 2048)
            assert np.array_equal(got, ref), (cell, opts)
    if mode == fr.FR_WINDING_I16:
        assert ref.max() == 10 and ref.min() == -10
    if mode == fr.FR_GRAY_DEBUG:
        assert ref.max() == 255 and ref.min() == 0
    # a 512 x 48 cell: two 256-pixel strips, three bands of 16 rows
    sg = synth_glyphset(2, 64, first_index=880)
    rows = [(i, int(np.floor(sg.boxes[i][0] * 0.25)), int(np.ceil(sg.boxes[i][3] * 0.25)) - 200, 512, 48, 0, 48 * i, np.float32(0.25)) for i in range(2)]
    got, ref = _batch_both(ctx, oracle, sg, rg.make_jobs(rows), mode, (96, 512), 1, False, threads=16)
    assert np.array_equal(got, ref)


def test_tall_cells_take_the_fast_kernels(ctx, oracle):
    """cells up to 512 pixels tall (16 samples) / 2048 (one sample) stay on cov4_kernel / win1_kernel: a 256 x 512
    and a 512 x 512 cell against the oracle, and a 128 x 1040 gray cell"""
    gs = synth_glyphset(2, 96, first_index=4040)
    rows = [(0, int(np.floor(gs.boxes[0][0] * 0.2)), int(np.ceil(gs.boxes[0][3] * 0.2)) + 40, 256, 512, 0, 0, np.float32(0.2)),
            (1, int(np.floor(gs.boxes[1][0] * 0.25)) - 8, int(np.ceil(gs.boxes[1][3] * 0.25)) + 8, 512, 512, 256, 0, np.float32(0.25))]
    jobs = rg.make_jobs(rows)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    plan = fr.Plan(dgs, jobs, fr.FR_COVERAGE_U8, 4, fr.FR_SAMPLE_CENTER)
    assert plan.stats() == {"jobs_cov4": 2, "jobs_general": 0}
    plan.close()
    got, ref = _batch_both(ctx, oracle, gs, jobs, fr.FR_COVERAGE_U8, (512, 768), 4, True, threads=16, dgs=dgs)
    assert np.array_equal(got, ref)
    tall = rg.make_jobs([(0, int(np.floor(gs.boxes[0][0] * 0.06)), int(np.ceil(gs.boxes[0][3] * 0.5)) + 8, 128, 1040, 0, 0, np.float32(0.5))])
    plan = fr.Plan(dgs, tall, fr.FR_GRAY_DEBUG, 1, fr.FR_SAMPLE_CORNER)
    assert plan.stats()["jobs_cov4"] == 1
    plan.close()
    got, ref = _batch_both(ctx, oracle, gs, tall, fr.FR_GRAY_DEBUG, (1040, 128), 1, False, threads=16, dgs=dgs)
    dgs.close()
    assert np.array_equal(got, ref) and (ref != 100).any()


def test_random_parity_sample():
    """a fixed-seed slice of tools/fuzz_parity.py (random glyph kinds, cell shapes incl. the cov4 / win1 shapes and mixed
    plans, scales, origins, unaligned destinations, all modes) as part of the suite; the tool itself has run 12 000+
    cases without a mismatch"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "4242", "150"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "150 cases, 0 mismatches" in out.stdout


def test_cov4_equals_general_kernel_on_2048_cells(ctx):
    """2 048 glyphs x 256^2 (a tenth of configs[2]) through cov4_kernel and through the general kernel:
    identical bytes (the general kernel is the one the oracle checks at small sizes)"""
    import torch
    G, cell, cols = 2048, 256, 32
    gs = synth_glyphset(G, 128, first_index=100000)
    jobs = cell_jobs(gs, cell, cell, 2048, cols)
    H, W = atlas_shape(G, cell, cols)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    a = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    b = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    plan = fr.Plan(dgs, jobs, fr.FR_COVERAGE_U8, 4, fr.FR_SAMPLE_CENTER)
    plan.render(a.data_ptr(), W, H)
    ctx.sync()
    plan.close()
    try:
        ctx.set_option("cov4", 0)
        plan = fr.Plan(dgs, jobs, fr.FR_COVERAGE_U8, 4, fr.FR_SAMPLE_CENTER)
        plan.render(b.data_ptr(), W, H)
        ctx.sync()
        plan.close()
    finally:
        ctx.set_option("cov4", 1)
    dgs.close()
    assert torch.equal(a, b)
    assert 0.2 < float((a > 0).float().mean()) < 0.8


def test_ragged_cells_unaligned_output(ctx, oracle, ascii_set):
    """odd sizes, unaligned destinations, a stride that is not a multiple of 16"""
    gs = ascii_set.gs
    rows = []
    rng = np.random.default_rng(3)
    x = 1
    for k, gi in enumerate(rng.integers(0, len(ascii_set), 12)):
        w, h = int(rng.integers(1, 70)), int(rng.integers(1, 300))
        s = np.float32(rng.integers(8, 120)) / np.float32(ascii_set.g_upm[gi])
        rows.append((int(gi), int(rng.integers(-20, 5)), int(rng.integers(20, 90)), w, h, x, int(rng.integers(0, 9)), s))
        x += w + int(rng.integers(0, 3))
    jobs = rg.make_jobs(rows)
    for mode, omode in MODES + [(fr.FR_COVERAGE_U8, O.COVERAGE_U8)]:
        n = 4 if mode == fr.FR_COVERAGE_U8 else 1
        got, ref = _batch_both(ctx, oracle, gs, jobs, mode, (311, x + 5), n, mode == fr.FR_COVERAGE_U8)
        assert np.array_equal(got, ref), mode


def test_overfull_rows_take_the_exact_fallback(ctx, oracle):
    """combs with 40 / 6 teeth: 80 / 12 crossings per ray.  The kernel keeps 8, 16 or 32 crossings
    per sample row in registers (kmax option); fuller rows take the exact direct-sum fallback."""
    gl = []
    for teeth in (40, 6):
        cs, box = comb_glyph(teeth)
        gl.append(Glyph(Box(*[int(v) for v in box]), [Contour(c) for c in cs]))
    gs = GlyphSet(gl)
    jobs = cell_jobs(gs, 192, 180, 2048, 2)
    try:
        for kmax in (32, 16, 8):
            ctx.set_option("kmax", kmax)
            for mode, n in [(fr.FR_COVERAGE_U8, 4), (fr.FR_WINDING_I16, 1), (fr.FR_COVERAGE_U8, 2), (fr.FR_COVERAGE_U8, 1), (fr.FR_GRAY_DEBUG, 1)]:
                got, ref = _batch_both(ctx, oracle, gs, jobs, mode, (192, 384), n, True)
                assert np.array_equal(got, ref), (kmax, mode, n)
        ctx.set_option("kmax", 8)
        sg = synth_glyphset(6, 48, first_index=900)
        sj = cell_jobs(sg, 80, 80, 2048, 3)
        got, ref = _batch_both(ctx, oracle, sg, sj, fr.FR_COVERAGE_U8, atlas_shape(6, 80, 3), 4, True)
        assert np.array_equal(got, ref)
        # the same through the uniform-plan instances (128- and 256-pixel cells), many over-full rows
        for cell, segs in ((128, 100), (256, 128)):
            sg = synth_glyphset(3, segs, first_index=7000 + cell)
            sj = cell_jobs(sg, cell, cell, 2048, 3)
            got, ref = _batch_both(ctx, oracle, sg, sj, fr.FR_COVERAGE_U8, atlas_shape(3, cell, 3), 4, True)
            assert np.array_equal(got, ref), cell
    finally:
        ctx.set_option("kmax", 32)


def test_wide_cells_are_split_into_strips(ctx, oracle, ascii_set):
    """a 700-pixel-wide render: column strips of 256 px (default) and of 48 px"""
    i = ascii_set.find("DejaVu", "W")
    g, upm = ascii_set.glyph(i), int(ascii_set.g_upm[i])
    gs = GlyphSet([g])
    try:
        for max_cols in (256, 48):
            ctx.set_option("strip_px", max_cols)
            jobs = cell_jobs(gs, 700, 600, upm, 1)
            jobs["h"] = 90
            jobs["max_y"] -= 200
            got, ref = _batch_both(ctx, oracle, gs, jobs, fr.FR_COVERAGE_U8, (90, 700), 4, True)
            assert np.array_equal(got, ref), max_cols
            got, ref = _batch_both(ctx, oracle, gs, jobs, fr.FR_GRAY_DEBUG, (90, 700))
            assert np.array_equal(got, ref), max_cols
    finally:
        ctx.set_option("strip_px", 256)


def test_edge_cases(ctx, oracle):
    # empty glyph: 1x1 image of value 100 (render_glyph.zig:14-22 with a zero box)
    im = fr.renderGlyph(Glyph.initEmpty(), fr.FontInformation(1000), 64, ctx=ctx)
    assert (im.width, im.height) == (1, 1) and im.data[0] == 100
    # a contour of one point has no curves (len/2 == 0, render_glyph.zig:38)
    g = Glyph(Box(0, 0, 10, 10), [Contour(np.array([[5, 5]], np.int16))])
    assert np.array_equal(fr.renderGlyph(g, fr.FontInformation(100), 50, ctx=ctx).as_2d(), oracle.render_glyph(g, 100, 50))
    # horizontal / vertical edges through sample rows, double roots, coincident control points
    pts = np.array([[0, 0], [0, 50], [0, 100], [50, 100], [100, 100], [100, 50], [100, 0], [50, 0], [0, 0]], np.int16)
    sq = Glyph(Box(0, 0, 100, 100), [Contour(pts)])
    for size in (100, 50, 7):
        assert np.array_equal(fr.renderGlyph(sq, fr.FontInformation(100), size, ctx=ctx).as_2d(), oracle.render_glyph(sq, 100, size))
    arch = Glyph(Box(0, 0, 200, 100), [Contour(np.array([[0, 0], [100, 200], [200, 0], [100, 0], [0, 0]], np.int16))])
    for size in (200, 64):   # rows tangent to the apex: delta == 0, the double root counted twice
        assert np.array_equal(fr.renderGlyph(arch, fr.FontInformation(200), size, ctx=ctx).as_2d(), oracle.render_glyph(arch, 200, size))
    # extreme coordinates of the i16 domain
    big = Glyph(Box(-32000, -32000, 32000, 32000), [Contour(np.array(
        [[-32000, -32000], [-32000, 0], [-32000, 32000], [0, 31000], [32000, 32000], [31000, 0], [32000, -32000], [0, -31000], [-32000, -32000]], np.int16))])
    assert np.array_equal(fr.renderGlyph(big, fr.FontInformation(64000), 97, ctx=ctx).as_2d(), oracle.render_glyph(big, 64000, 97))
    # no jobs / bad arguments
    dgs = fr.DeviceGlyphSet(ctx, GlyphSet([sq]))
    with pytest.raises(fr.FrError):
        rg.render_batch(dgs, rg.make_jobs([(5, 0, 0, 4, 4, 0, 0, 1.0)]), fr.FR_GRAY_DEBUG, np.zeros((4, 4), np.uint8))
    with pytest.raises(fr.FrError):
        rg.render_batch(dgs, rg.make_jobs([(0, 0, 0, 8, 8, 0, 0, 1.0)]), fr.FR_GRAY_DEBUG, np.zeros((4, 4), np.uint8))
    with pytest.raises(fr.FrError):
        fr.DeviceGlyphSet(ctx, GlyphSet.from_arrays(np.zeros((4, 2), np.int16), [0, 4], [0, 1], [[0, 0, 1, 1]]))   # even length
    dgs.close()


def test_random_scales_and_offsets_property(ctx, oracle):
    """many (scale, origin) pairs on one synthetic glyph: rows landing exactly on vertices etc."""
    gs = synth_glyphset(1, 40, first_index=4242)
    rng = np.random.default_rng(11)
    rows, y = [], 0
    for k in range(24):
        s = np.float32(rng.integers(3, 400)) / np.float32(2048)
        rows.append((0, int(rng.integers(-3, 40)), int(rng.integers(10, 400)), 48, 40, (k % 6) * 48, (k // 6) * 40, s))
    jobs = rg.make_jobs(rows)
    for mode in (fr.FR_WINDING_I16, fr.FR_COVERAGE_U8):
        got, ref = _batch_both(ctx, oracle, gs, jobs, mode, (160, 288), 4 if mode == fr.FR_COVERAGE_U8 else 1, False)
        assert np.array_equal(got, ref)


def test_zoomed_and_shrunk_cells(ctx, oracle, ascii_set):
    """scales far from the usual size/upm: cells that look at a small part of an outline at 37.5, 1000 and
    4096 pixels per font unit (pixel coordinates up to 2^22: the affine row guess is off by several rows
    and the settle loops walk), and whole glyphs squeezed into a few pixels"""
    i = ascii_set.find("STIX", "g")
    gs = GlyphSet([ascii_set.glyph(i)])
    box = gs.boxes[0].astype(np.int64)
    rows, x = [], 0
    for s, (fx, fy) in [(37.5, (0.5, 0.5)), (1000.0, (0.3, 0.6)), (4096.0, (0.52, 0.41)), (333.25, (0.1, 0.9))]:
        ux, uy = box[0] + fx * (box[2] - box[0]), box[1] + fy * (box[3] - box[1])      # font-unit point to look at
        rows.append((0, int(ux * s) - 30, int(uy * s) + 20, 61, 43, x, 0, np.float32(s)))
        x += 64
    for s in (0.004, 0.0009765625, 0.02):
        rows.append((0, int(np.floor(box[0] * s)) - 2, int(np.ceil(box[3] * s)) + 2, 23, 19, x, 0, np.float32(s)))
        x += 32
    jobs = rg.make_jobs(rows)
    for mode, n, center in [(fr.FR_COVERAGE_U8, 4, True), (fr.FR_COVERAGE_U8, 4, False), (fr.FR_WINDING_I16, 1, False), (fr.FR_COVERAGE_U8, 2, True)]:
        got, ref = _batch_both(ctx, oracle, gs, jobs, mode, (48, x), n, center)
        assert np.array_equal(got, ref), (mode, n, center)
        assert got.any()


def test_uniform_plan_instances_equal_the_general_kernel(ctx, oracle):
    """plans whose every job is a whole number of 256- / 128-pixel strips wide and of wave bands tall run
    kernel instances with compile-time shapes; one odd-sized job in the batch sends the same cells through
    the general instance — identical bytes, and both equal the oracle"""
    for cell, segs in ((256, 96), (128, 40)):
        gs = synth_glyphset(5, segs, first_index=3100 + cell)
        jobs_u = cell_jobs(gs, cell, cell, 2048, 4)[:4]                      # uniform: 4 full cells
        jobs_g = cell_jobs(gs, cell, cell, 2048, 5).copy()                   # + a fifth, 3 pixels short
        jobs_g["w"][4] = cell - 3
        jobs_g["h"][4] = cell - 3
        dgs = fr.DeviceGlyphSet(ctx, gs)
        a = np.zeros((cell, 4 * cell), np.uint8)
        b = np.zeros((cell, 5 * cell), np.uint8)
        rg.render_batch(dgs, jobs_u, fr.FR_COVERAGE_U8, a, 4, fr.FR_SAMPLE_CENTER)
        rg.render_batch(dgs, jobs_g, fr.FR_COVERAGE_U8, b, 4, fr.FR_SAMPLE_CENTER)
        dgs.close()
        assert np.array_equal(a, b[:, :4 * cell]), cell
        ref = np.zeros_like(b)
        oracle.render_batch(gs, jobs_g, O.COVERAGE_U8, ref, 4, True, 8)
        assert np.array_equal(b, ref), cell


def test_exact_integer_path(ctx, oracle, ascii_set):
    """GlyphInfo.init + windingInGlyph (render_glyph.zig:76-300) incl. the GlyphDebug lattice"""
    oracle.diag_reset()
    for font, ch in [("STIX", "A"), ("STIX", "g"), ("DejaVu", "S"), ("DejaVu", "8"), ("STIX", "@")]:
        g = ascii_set.glyph(ascii_set.find(font, ch))
        info = fr.GlyphInfo.init(g, ctx=ctx)
        ct, ip = oracle.glyph_info(g)
        assert np.array_equal(info.curve_type, ct) and np.array_equal(info.include_p0, ip)
        assert np.array_equal(fr.winding_lattice(g, ctx=ctx), oracle.winding_lattice(g)), (font, ch)
    g = ascii_set.glyph(ascii_set.find("STIX", "A"))
    q = np.array([(420, 321), (14, 0), (360, 674), (100, 0), (-5, -5)], np.int16)
    assert np.array_equal(fr.windingInGlyph(g, None, q, ctx=ctx), oracle.winding_in_glyph(g, q))
    assert fr.windingInGlyph(g, None, (420, 321), ctx=ctx) == 0
    assert oracle.diag()[0] == 0          # fixtures stay inside the reference's i64 domain
    # synthetic long curves: beyond i64 the reference is undefined; both sides use 128-bit
    gs = synth_glyphset(1, 24, first_index=77)
    gg = gs.glyph(0)
    qq = np.stack(np.meshgrid(np.arange(60, 2000, 37), np.arange(60, 2000, 41)), -1).reshape(-1, 2).astype(np.int16)
    assert np.array_equal(fr.windingInGlyph(gg, None, qq, ctx=ctx), oracle.winding_in_glyph(gg, qq))


def test_exact_lattice_k_and_coverage(ctx, oracle, ascii_set):
    """SURVEY §8 f-3 (build-defined): a4-a7 on the glyph scaled by K at integer lattice points, and
    the n x n box-filtered coverage built on it — bit-exact vs the oracle's twin; K = 1 with the
    GlyphDebug origin is the reference's own lattice"""
    for font, ch in [("STIX", "A"), ("DejaVu", "8"), ("STIX", "@")]:
        g = ascii_set.glyph(ascii_set.find(font, ch))
        box = g.box.as_array().astype(int)
        W, H = box[2] - box[0] + 3, box[3] - box[1] + 3
        assert np.array_equal(fr.exact_lattice(g, 1, box[0] - 1, box[3] + 1, W, H, ctx=ctx), fr.winding_lattice(g, ctx=ctx))
        for K in (2, 3, 8):
            # a window of the K-refined lattice around the top-left quarter of the glyph
            x0, y0 = K * (box[0] - 2), K * (box[3] + 2)
            w, h = 97, 83
            assert np.array_equal(fr.exact_lattice(g, K, x0, y0, w, h, ctx=ctx), oracle.exact_lattice(g, K, x0, y0, w, h)), (font, ch, K)
        # 4 x 4 lattice points per pixel at 1/8 font unit spacing (K = 8), 40 x 30 pixels over the glyph's centre
        K, n = 8, 4
        cx, cy = (box[0] + box[2]) // 2, (box[1] + box[3]) // 2
        got = fr.exact_coverage(g, K, K * cx - 80, K * cy + 60, 40, 30, n, ctx=ctx)
        assert np.array_equal(got, oracle.exact_coverage(g, K, K * cx - 80, K * cy + 60, 40, 30, n)), (font, ch)
    # a coarse whole-glyph coverage: K = 1, n = 4 -> one pixel per 4 x 4 font units
    g = ascii_set.glyph(ascii_set.find("STIX", "g"))
    box = g.box.as_array().astype(int)
    wp, hp = (box[2] - box[0]) // 4 + 2, (box[3] - box[1]) // 4 + 2
    got = fr.exact_coverage(g, 1, box[0] - 2, box[3] + 2, wp, hp, 4, ctx=ctx)
    assert np.array_equal(got, oracle.exact_coverage(g, 1, box[0] - 2, box[3] + 2, wp, hp, 4))
    assert got.max() == 255 and got.min() == 0
    with pytest.raises(fr.FrError):
        fr.exact_lattice(g, 9, 0, 0, 4, 4, ctx=ctx)          # K > 8 could leave the 128-bit range


def test_full_size_properties_config3(ctx):
    """BASELINE configs[2] at FULL size (20 992 glyphs x 256^2, S = 128, 16 samples/pixel, 1.376
    Gpixel): too big for the oracle, so size-independent properties — re-render is idempotent,
    8 shards rendered one after another == the unsharded atlas byte for byte (SURVEY §4 item 5:
    the single-GPU "fake cluster"), a 64-glyph prefix equals the oracle-checked small render,
    and the inside fraction is sane.  Everything stays on the device (torch.equal)."""
    import torch
    from font_renderer_amd.shard import shard_ranges
    G, cols, cell = 20992, 64, 256
    gs = synth_glyphset(G, 128)
    H, W = atlas_shape(G, cell, cols)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    jobs = cell_jobs(gs, cell, cell, 2048, cols)
    out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    plan = fr.Plan(dgs, jobs, fr.FR_COVERAGE_U8, 4, fr.FR_SAMPLE_CENTER)
    assert plan.pixels == G * cell * cell
    plan.render(out.data_ptr(), W, H); ctx.sync()
    first = out.clone()
    dgs.prepare(); plan.render(out.data_ptr(), W, H); ctx.sync()
    assert torch.equal(out, first)                                   # idempotent (prepare + render)
    out2 = torch.zeros_like(out)
    for lo, hi in shard_ranges(G, 8):                                # 2 624 glyphs = 41 atlas rows per shard
        sub = gs.subset(lo, hi)
        sd = fr.DeviceGlyphSet(ctx, sub)
        band = out2[(lo // cols) * cell:]
        sp = fr.Plan(sd, cell_jobs(sub, cell, cell, 2048, cols), fr.FR_COVERAGE_U8, 4, fr.FR_SAMPLE_CENTER)
        sp.render(band.data_ptr(), W, band.shape[0]); ctx.sync()
        sp.close(); sd.close()
    assert torch.equal(out2, first)
    frac = float((first[: 8 * cell] > 0).float().mean())
    assert 0.2 < frac < 0.8
    # the first atlas row (64 glyphs) against a separately rendered small batch
    small = torch.zeros((cell, W), dtype=torch.uint8, device="cuda")
    sub = gs.subset(0, cols)
    sd = fr.DeviceGlyphSet(ctx, sub)
    sp = fr.Plan(sd, cell_jobs(sub, cell, cell, 2048, cols), fr.FR_COVERAGE_U8, 4, fr.FR_SAMPLE_CENTER)
    sp.render(small.data_ptr(), W, cell); ctx.sync()
    assert torch.equal(small, first[:cell])
    sp.close(); sd.close(); plan.close(); dgs.close()


def test_full_size_properties_config4_shard(ctx, oracle):
    """BASELINE configs[3] per-GPU share at FULL size (63 488 / 8 = 7 936 glyphs x 128^2, S = 32, 16 samples per
    pixel): re-render idempotent, 8 sub-shards rendered one after another == the unsharded atlas, the general
    kernel (ctx option cov4 = 0) gives the same bytes as cov4_kernel, and 6 cells against the oracle."""
    import torch
    from font_renderer_amd.shard import shard_ranges
    G, cols, cell = 7936, 64, 128
    gs = synth_glyphset(G, 32)
    H, W = atlas_shape(G, cell, cols)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    jobs = cell_jobs(gs, cell, cell, 2048, cols)
    out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    plan = fr.Plan(dgs, jobs, fr.FR_COVERAGE_U8, 4, fr.FR_SAMPLE_CENTER)
    assert plan.pixels == G * cell * cell and plan.stats() == {"jobs_cov4": G, "jobs_general": 0}
    plan.render(out.data_ptr(), W, H); ctx.sync()
    first = out.clone()
    plan.render(out.data_ptr(), W, H); ctx.sync()
    assert torch.equal(out, first)
    plan.close()
    out2 = torch.zeros_like(out)
    torch.cuda.synchronize()
    for lo, hi in shard_ranges(G, 8):                                # 992 glyphs = 15.5 atlas rows: ragged bands
        sub = gs.subset(lo, hi)
        sd = fr.DeviceGlyphSet(ctx, sub)
        sj = cell_jobs(sub, cell, cell, 2048, cols).copy()
        sj["out_x"] = ((np.arange(lo, hi) % cols) * cell).astype(np.uint32)      # keep every glyph in its unsharded cell
        sj["out_y"] = ((np.arange(lo, hi) // cols) * cell).astype(np.uint32)
        sp = fr.Plan(sd, sj, fr.FR_COVERAGE_U8, 4, fr.FR_SAMPLE_CENTER)
        sp.render(out2.data_ptr(), W, H); ctx.sync()
        sp.close(); sd.close()
    assert torch.equal(out2, first)
    try:
        ctx.set_option("cov4", 0)
        gp = fr.Plan(dgs, jobs, fr.FR_COVERAGE_U8, 4, fr.FR_SAMPLE_CENTER)
        assert gp.stats()["jobs_cov4"] == 0
        out2.zero_(); torch.cuda.synchronize()
        gp.render(out2.data_ptr(), W, H); ctx.sync()
        gp.close()
    finally:
        ctx.set_option("cov4", 1)
    assert torch.equal(out2, first)
    dgs.close()
    host = first.cpu().numpy()
    for gi in (0, 63, 64, 3000, 7000, G - 1):
        g1 = synth_glyphset(1, 32, first_index=gi)
        ref = np.zeros((cell, cell), np.uint8)
        oracle.render_batch(g1, cell_jobs(g1, cell, cell, 2048, 1), O.COVERAGE_U8, ref, 4, True)
        y, x = (gi // cols) * cell, (gi % cols) * cell
        assert np.array_equal(host[y:y + cell, x:x + cell], ref), gi


def test_sdf_config5_cells_and_exact_culls(ctx, oracle):
    """BASELINE configs[4] shape: SDF at 512 x 512 per glyph (two 256-pixel strips per cell) on 3 synthetic S = 64
    glyphs against the CPU twin; and the claim behind the kernel's culls (a region and a quad drop segments beyond the
    encoding's reach, a pixel skips segments no nearer than its best so far): with the culls switched off (ctx
    option sdf_cull = 0) the bytes are the same — on the 512^2 cells and on 1 500 small random cells."""
    gs = synth_glyphset(3, 64, first_index=5151)
    jobs = cell_jobs(gs, 512, 512, 2048, 3)
    got, ref = _batch_both(ctx, oracle, gs, jobs, fr.FR_SDF_U8, (512, 1536), 1, True, threads=16)
    assert np.array_equal(got, ref)
    assert got.min() == 0 and got.max() == 255
    try:
        ctx.set_option("sdf_cull", 0)
        dgs = fr.DeviceGlyphSet(ctx, gs)
        plain = np.zeros_like(got)
        rg.render_batch(dgs, jobs, fr.FR_SDF_U8, plain, 1, fr.FR_SAMPLE_CENTER)
        dgs.close()
    finally:
        ctx.set_option("sdf_cull", 1)
    assert np.array_equal(plain, got)
    # many small cells at random scales and origins (cull decisions at every distance from the outline)
    gs = synth_glyphset(30, 48, first_index=9300)
    rng = np.random.default_rng(77)
    rows = []
    for k in range(1500):
        gi = int(rng.integers(0, 30))
        s = np.float32(rng.integers(16, 700)) / np.float32(2048)
        rows.append((gi, int(rng.integers(-40, int(2048 * s))), int(rng.integers(0, int(2048 * s) + 40)), 32, 32, (k % 50) * 32, (k // 50) * 32, s))
    jobs = rg.make_jobs(rows)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    a = np.zeros((30 * 32, 50 * 32), np.uint8)
    b = np.zeros_like(a)
    rg.render_batch(dgs, jobs, fr.FR_SDF_U8, a, 1, fr.FR_SAMPLE_CORNER)
    try:
        ctx.set_option("sdf_cull", 0)
        rg.render_batch(dgs, jobs, fr.FR_SDF_U8, b, 1, fr.FR_SAMPLE_CORNER)
    finally:
        ctx.set_option("sdf_cull", 1)
    dgs.close()
    assert np.array_equal(a, b)
    assert 0 < (a == 0).mean() < 1 and 0 < (a == 255).mean() < 1


def test_glyph_debug_render_and_atlas_page_to_qoi(ctx, oracle, ascii_set):
    """SURVEY §8 f-4 / f-2 in the product: Image.GlyphDebug.render (Image.zig:173-241) through
    fr_glyph_debug_render == the oracle's restatement; a GPU-rendered 2048^2 atlas page (configs[1]: 95 ASCII
    glyphs, 128-pixel cells, laid out by fr_atlas_layout) -> fr_qoi_encode_gray -> decoded by Pillow == the page,
    and the stream == the oracle's encoder on the oracle's page."""
    import io
    from PIL import Image as PILImage
    from font_renderer_amd import qoi
    for font, ch, ws in (("STIX", "i", 50), ("DejaVu", "g", 200), ("STIX", "A", 255)):
        g = ascii_set.glyph(ascii_set.find(font, ch))
        dbg = fr.GlyphDebug.render(g, ws, ctx=ctx)
        want = oracle.glyph_debug_render(g, ws)
        assert (dbg.rgb.height, dbg.rgb.width) == want.shape[:2]
        assert np.array_equal(dbg.rgb.as_3d(), want), (font, ch)
    gs = ascii_set.gs
    jobs, pages, n_pages = cell_jobs(gs, 128, 100, ascii_set.g_upm, 16, first_glyph=0, n_glyphs=95, page_rows=16, return_pages=True)
    assert n_pages == 1 and (pages == 0).all()
    got, ref = _batch_both(ctx, oracle, gs, jobs, fr.FR_COVERAGE_U8, (2048, 2048), 4, True)
    got[got == 7] = 0; ref[ref == 7] = 0          # (cells past the 95th keep the canvas value of _batch_both)
    assert np.array_equal(got, ref)
    stream = qoi.saveRGB(got)
    dec = np.asarray(PILImage.open(io.BytesIO(stream)).convert("RGB"))
    assert np.array_equal(dec[:, :, 0], got) and np.array_equal(dec[:, :, 1], got) and np.array_equal(dec[:, :, 2], got)
    assert stream == oracle.qoi_encode(np.repeat(ref[:, :, None], 3, 2))


def test_two_contexts_two_streams_one_thread(ctx, oracle):
    """the shape a single-threaded Zig host would give a multi-stream job: two fr_ctx (own streams) driven
    alternately from one thread, renders in flight on both; each result equals the oracle"""
    import torch
    gs_a, gs_b = synth_glyphset(24, 64, first_index=70), stroke_glyphset(24, 96, first_index=71)
    cell, cols = 128, 8
    with fr.Context(0) as ca, fr.Context(0) as cb:
        da, db = fr.DeviceGlyphSet(ca, gs_a), fr.DeviceGlyphSet(cb, gs_b)
        ja, jb = cell_jobs(gs_a, cell, cell, 2048, cols), cell_jobs(gs_b, cell, cell, 2048, cols)
        pa = fr.Plan(da, ja, fr.FR_COVERAGE_U8, 4, fr.FR_SAMPLE_CENTER)
        pb = fr.Plan(db, jb, fr.FR_COVERAGE_U8, 4, fr.FR_SAMPLE_CENTER)
        H, W = atlas_shape(24, cell, cols)
        oa = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
        ob = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for _ in range(3):                         # interleaved launches, nothing waited for in between
            pa.render(oa.data_ptr(), W, H)
            pb.render(ob.data_ptr(), W, H)
        ca.sync(); cb.sync()
        pa.close(); pb.close(); da.close(); db.close()
    for gsx, jx, ox in ((gs_a, ja, oa), (gs_b, jb, ob)):
        ref = np.zeros((H, W), np.uint8)
        oracle.render_batch(gsx, jx, O.COVERAGE_U8, ref, 4, True, 16)
        assert np.array_equal(ox.cpu().numpy(), ref)


def test_oracle_spot_check_inside_full_size_workload(ctx, oracle):
    """4 cells of the C3 workload (first, two middle, last glyph index) against the oracle."""
    G = 20992
    for gi in (0, 7777, 15000, G - 1):
        gs = synth_glyphset(1, 128, first_index=gi)
        jobs = cell_jobs(gs, 256, 256, 2048, 1)
        got, ref = _batch_both(ctx, oracle, gs, jobs, fr.FR_COVERAGE_U8, (256, 256), 4, True, threads=16)
        assert np.array_equal(got, ref), gi


def test_cpp_host_mirror(ctx, ascii_set, tmp_path):
    """font-renderer_amd/host/fr_host.hpp (C++ mirror of renderGlyph / Image.Gray / windingInGlyph)
    through the same C ABI: STIX 'A' at 64 -> 47x45, Appendix B histogram, same bytes as Python."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "font-renderer_amd", "host", "host_selftest")
    i = ascii_set.find("STIX", "A")
    gs1 = ascii_set.gs.subset(i, i + 1)
    gs1.points_xy.astype("<i2").tofile(tmp_path / "pts.bin")
    gs1.contour_start.astype("<u4").tofile(tmp_path / "cs.bin")
    env = dict(os.environ, FR_HIP_RUNTIME="system")
    out = subprocess.run([exe, str(tmp_path / "pts.bin"), str(tmp_path / "cs.bin")], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    w, h, hm2, hm1, h0, h1, fnv, wi, fnv_dbg, fnv_atlas = out.stdout.split()
    assert (int(w), int(h), int(hm2), int(hm1), int(h0), int(h1)) == (47, 45, 1, 28, 1641, 445)
    assert int(wi) == 0
    gray = fr.renderGlyph(ascii_set.glyph(i), fr.FontInformation(1000), 64, ctx=ctx).data
    hh = 1469598103934665603
    for b in gray.tolist():
        hh = ((hh ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert int(fnv, 16) == hh

    def fnv1a(buf):
        x = 1469598103934665603
        for b in buf.tolist():
            x = ((x ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return x
    g = ascii_set.glyph(i)
    assert int(fnv_dbg, 16) == fnv1a(fr.GlyphDebug.render(g, 50, ctx=ctx).rgb.data.reshape(-1))
    gs2 = GlyphSet([g, g])
    atlas = np.zeros((128, 256), np.uint8)
    dgs = fr.DeviceGlyphSet(ctx, gs2)
    rg.render_batch(dgs, cell_jobs(gs2, 128, 100, 1000, 2), fr.FR_COVERAGE_U8, atlas, 4, fr.FR_SAMPLE_CENTER)
    dgs.close()
    assert int(fnv_atlas, 16) == fnv1a(atlas.reshape(-1))


def test_division_shortcut_is_exact_exhaustively(ctx):
    """div_by_int (reciprocal multiply + Markstein FMA correction) == IEEE x / d for ALL 2^23
    significands x every integer divisor 1..2^17, both signs, three binades — run on the device.
    This is the proof obligation for using it in place of render_glyph.zig:51,60-61's `/`."""
    import ctypes as C
    from font_renderer_amd import _lib
    lib = _lib.load_library()
    m, bd, bx = C.c_uint64(1), C.c_uint32(), C.c_uint32()
    _lib.check(lib.fr_selftest_division(1, 1 << 17, C.byref(m), C.byref(bd), C.byref(bx)))
    assert m.value == 0, f"{m.value} mismatches, e.g. divisor {bd.value}, x bits {bx.value:#x}"


def test_sqrt_shortcut_is_exact_exhaustively(ctx):
    """sqrt_rn (v_sqrt_f32 + two FMA residuals, no denormal pre-scaling) == the correctly rounded
    sqrt for EVERY binary32 in [2^-30, 2^66) — run on the device.  Proof obligation for using it
    for render_glyph.zig:60's std.math.sqrt; delta is 0, negative or >= 2^-24."""
    import ctypes as C
    from font_renderer_amd import _lib
    lib = _lib.load_library()
    m, bx = C.c_uint64(1), C.c_uint32()
    _lib.check(lib.fr_selftest_sqrt(C.byref(m), C.byref(bx)))
    assert m.value == 0, f"{m.value} mismatches, e.g. x bits {bx.value:#x}"


def test_whole_font_from_the_c_side_producer(ctx, oracle):
    """SURVEY §8f-1 end to end: TrueType bytes -> fr_font_* (C++ contour producer) -> glyph set ->
    one atlas of every glyph the reference could load (DejaVuSerif-Italic: ~3 000 glyphs incl.
    composites), 16 samples/pixel; 48 random cells against the oracle."""
    import os
    path = "/usr/local/lib/python3.10/dist-packages/matplotlib/mpl-data/fonts/ttf/DejaVuSerif-Italic.ttf"
    if not os.path.exists(path):
        pytest.skip("font not present")
    font = fr.Font.initTTF(path)
    gs, kept = font.glyphset()
    assert len(kept) > 3000
    cell, cols = 48, 64
    jobs = cell_jobs(gs, cell, 40, font.information.units_per_em, cols)
    H, W = atlas_shape(len(gs), cell, cols)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    atlas = np.zeros((H, W), np.uint8)
    rg.render_batch(dgs, jobs, fr.FR_COVERAGE_U8, atlas, 4, fr.FR_SAMPLE_CENTER)
    dgs.close()
    rng = np.random.default_rng(9)
    for k in rng.choice(len(gs), 48, replace=False):
        sub = gs.subset(int(k), int(k) + 1)
        j1 = cell_jobs(sub, cell, 40, font.information.units_per_em, 1)
        ref = np.zeros((cell, cell), np.uint8)
        oracle.render_batch(sub, j1, O.COVERAGE_U8, ref, 4, True)
        y, x = (int(k) // cols) * cell, (int(k) % cols) * cell
        assert np.array_equal(atlas[y:y + cell, x:x + cell], ref), int(k)
    assert (atlas > 0).mean() > 0.02


@pytest.mark.parametrize("center", [False, True])
def test_sdf_matches_its_cpu_twin(ctx, oracle, ascii_set, center):
    """FR_SDF_U8 (BASELINE configs[4]; build-defined, the reference has no SDF): GPU == the C twin in
    the oracle bit for bit — real glyphs at 64x64 and synthetic S=64 glyphs at 96x96 (config 5 shape
    at reduced cell size: the CPU twin is O(pixels x segments))."""
    gs = ascii_set.gs
    jobs = cell_jobs(gs, 64, 48, ascii_set.g_upm, 8, first_glyph=33, n_glyphs=24)
    got, ref = _batch_both(ctx, oracle, gs, jobs, fr.FR_SDF_U8, (3 * 64, 8 * 64), 1, center, threads=16)
    assert np.array_equal(got, ref)
    assert got.min() == 0 and got.max() > 140
    # the empty glyph (' ': no segment, every byte 0) next to its neighbours, in cells smaller than one 32-pixel region row
    i0 = ascii_set.find("STIX", " ")
    ej = cell_jobs(gs, 40, 30, ascii_set.g_upm, 4, first_glyph=i0, n_glyphs=4)
    got0, ref0 = _batch_both(ctx, oracle, gs, ej, fr.FR_SDF_U8, (40, 160), 1, center, threads=4)
    assert np.array_equal(got0, ref0) and not got0[:, :40].any()
    sg = synth_glyphset(6, 64, first_index=5000)
    sj = cell_jobs(sg, 96, 96, 2048, 3)
    got, ref = _batch_both(ctx, oracle, sg, sj, fr.FR_SDF_U8, atlas_shape(6, 96, 3), 1, center, threads=16)
    assert np.array_equal(got, ref)
    # larger cells: most 16x16 tiles lie beyond the encoding's 8-pixel reach of most segments (the
    # kernel drops those while staging) and many saturate outright
    sg2 = synth_glyphset(2, 64, first_index=5100)
    sj2 = cell_jobs(sg2, 224, 224, 2048, 2)
    got2, ref2 = _batch_both(ctx, oracle, sg2, sj2, fr.FR_SDF_U8, atlas_shape(2, 224, 2), 1, center, threads=16)
    assert np.array_equal(got2, ref2)
    assert (got2 == 0).any() and (got2 == 255).any()
    # glyphs of more than 64 segments (the kernel reads 64 at a time and parks the minima in LDS in between), in cells
    # whose sides are no multiple of its 8-pixel quads or 32-pixel regions; culls on == culls off == the twin
    for big, cw, ch in ((synth_glyphset(3, 200, first_index=5200), 88, 72), (stroke_glyphset(2, 256, first_index=5300), 100, 117)):
        sc = np.float32(min(cw, ch) - 8) / np.float32(2048)
        bj = rg.make_jobs([(g, -5, ch - 4, cw, ch, g * cw, 0, sc) for g in range(len(big))])
        got3, ref3 = _batch_both(ctx, oracle, big, bj, fr.FR_SDF_U8, (ch, cw * len(big)), 1, center, threads=16)
        assert np.array_equal(got3, ref3)
        try:
            ctx.set_option("sdf_cull", 0)
            plain = np.zeros_like(got3)
            dg = fr.DeviceGlyphSet(ctx, big)
            rg.render_batch(dg, bj, fr.FR_SDF_U8, plain, 1, fr.FR_SAMPLE_CENTER if center else fr.FR_SAMPLE_CORNER)
            dg.close()
        finally:
            ctx.set_option("sdf_cull", 1)
        assert np.array_equal(plain, ref3)
    # sign agrees with the coverage mask: inside -> >= 128, outside -> <= 128 (|d| < 1/32 px rounds to 128)
    mask = np.zeros_like(got)
    rg.render_batch(fr.DeviceGlyphSet(ctx, sg), sj, fr.FR_MASK_NONZERO, mask, 1, fr.FR_SAMPLE_CENTER if center else fr.FR_SAMPLE_CORNER)
    assert (got[mask == 255] >= 128).all() and (got[mask == 0] <= 128).all()
