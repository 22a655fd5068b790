"""GPU suite, round 3: the reference's own product shape in batch.

renderGlyph sizes every image to the glyph's own box (render_glyph.zig:14-19: 47 x 45 for STIX 'A' at 64).  These tests
put such ragged cells — any width, any height, unaligned destinations — through the fast kernels (cov4_kernel with
4 x 4 and 2 x 2 samples, win1_kernel with one) and compare with the oracle byte for byte, including the bytes AROUND
every cell (the clipped stores must not touch them) and the kernel split the plan reports."""
import numpy as np
import pytest

import font_renderer_amd as fr
import oracle_lib as O
from font_renderer_amd import render_glyph as rg
from font_renderer_amd.atlas import atlas_shape, cell_jobs, glyph_dims_jobs
from font_renderer_amd.glyph import Box, Contour, Glyph, GlyphSet
from font_renderer_amd.synth import comb_glyph, stroke_glyphset, synth_glyphset

pytestmark = pytest.mark.gpu

ONE = [(fr.FR_WINDING_I16, O.WINDING_I16, 1, False), (fr.FR_GRAY_DEBUG, O.GRAY_DEBUG, 1, False), (fr.FR_MASK_NONZERO, O.MASK_NONZERO, 1, False)]
COV = [(fr.FR_COVERAGE_U8, O.COVERAGE_U8, n, c) for n in (1, 2, 4) for c in (False, True)]


def _both(ctx, oracle, gs, jobs, mode, omode, shape, n, center, dgs, expect_general=0, want_in_describe=()):
    dt = np.int16 if mode == fr.FR_WINDING_I16 else np.uint8
    got = np.full(shape, 0x5b, dt)
    ref = np.full(shape, 0x5b, dt)
    phase = fr.FR_SAMPLE_CENTER if center else fr.FR_SAMPLE_CORNER
    plan = fr.Plan(dgs, jobs, mode, n, phase)
    st, desc = plan.stats(), plan.describe()
    plan.close()
    assert st["jobs_general"] == expect_general and st["jobs_cov4"] == len(jobs) - expect_general, (st, desc)
    for w in want_in_describe:
        assert w in desc, (w, desc)
    rg.render_batch(dgs, jobs, mode, got, n, phase)
    oracle.render_batch(gs, jobs, omode, ref, n, center, 16)
    return got, ref, desc


@pytest.mark.parametrize("font_size", [24, 64, 150])
def test_renderglyph_dims_batch_takes_the_fast_kernels(ctx, oracle, ascii_set, font_size):
    """190 real glyphs, each at exactly renderGlyph's image size (render_glyph.zig:13-19), shelf-packed tight (unaligned
    out_x, a stride that is no multiple of 16): every mode == the oracle, nothing on the general kernel, the 64-pixel
    strip instance in use for the small sizes; the gray job bytes == what fr_render_glyph returns for the glyph."""
    gs = ascii_set.gs
    jobs, H = glyph_dims_jobs(gs, font_size, ascii_set.g_upm, 1021)
    assert len(jobs) == len(gs) and H > 0
    i = ascii_set.find("STIX", "A")
    if font_size == 64:
        assert (int(jobs["w"][i]), int(jobs["h"][i])) == (47, 45)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    for mode, omode, n, center in ONE + COV:
        fast = "win1_kernel" if n == 1 else "cov4_kernel"
        # (150: widths of 40 .. 150 pixels — classes of fewer than 64 jobs join the next one up, the widest always exists)
        want = [f"fr::{fast}<2, "] if font_size <= 64 else [f"fr::{fast}<4, "]
        got, ref, desc = _both(ctx, oracle, gs, jobs, mode, omode, (H + 3, 1021), n, center, dgs, 0, want)
        assert np.array_equal(got, ref), (mode, n, center, desc)
        if mode == fr.FR_GRAY_DEBUG:
            assert (ref != 0x5b).any()
            for k in (i, 3, 77, 150):
                g = ascii_set.glyph(k)
                im = fr.renderGlyph(g, fr.FontInformation(int(ascii_set.g_upm[k])), font_size, ctx=ctx)
                j = jobs[k]
                assert (im.width, im.height) == (int(j["w"]), int(j["h"]))
                assert np.array_equal(im.as_2d(), got[j["out_y"]:j["out_y"] + j["h"], j["out_x"]:j["out_x"] + j["w"]])
    dgs.close()


def test_ragged_cells_unaligned_output_on_the_fast_kernels(ctx, oracle, ascii_set):
    """the generator of test_ragged_cells_unaligned_output (odd sizes 1..69 x 1..299, unaligned destinations, arbitrary
    origins and scales), now asserting the split: every job on cov4_kernel / win1_kernel"""
    gs = ascii_set.gs
    rows = []
    rng = np.random.default_rng(3)
    x = 1
    for gi in rng.integers(0, len(ascii_set), 12):
        w, h = int(rng.integers(1, 70)), int(rng.integers(1, 300))
        s = np.float32(rng.integers(8, 120)) / np.float32(ascii_set.g_upm[gi])
        rows.append((int(gi), int(rng.integers(-20, 5)), int(rng.integers(20, 90)), w, h, x, int(rng.integers(0, 9)), s))
        x += w + int(rng.integers(0, 3))
    # + widths around the strip widths and heights around the band heights
    for w, h in ((63, 15), (64, 16), (65, 17), (127, 31), (128, 33), (129, 47), (255, 48), (256, 49), (257, 5), (300, 70), (16, 1), (1, 64)):
        gi = int(rng.integers(0, len(ascii_set)))
        s = np.float32(rng.integers(30, 300)) / np.float32(ascii_set.g_upm[gi])
        rows.append((gi, int(rng.integers(-10, 5)), int(rng.integers(10, 200)), w, h, x, int(rng.integers(0, 5)), s))
        x += w + int(rng.integers(0, 3))
    jobs = rg.make_jobs(rows)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    for mode, omode, n, center in ONE + COV:
        got, ref, desc = _both(ctx, oracle, gs, jobs, mode, omode, (311, x + 5), n, center, dgs)
        assert np.array_equal(got, ref), (mode, n, center, desc)
    dgs.close()


@pytest.mark.parametrize("segs,cell", [(32, 128), (128, 256), (200, 96), (64, 50)])
def test_two_by_two_samples_on_cov4(ctx, oracle, segs, cell):
    """n = 2 (4 samples per pixel) on cov4_kernel<.., 2>: synthetic and stroke-dense glyphs, both phases"""
    for make, first in ((synth_glyphset, 700), (stroke_glyphset, 900)):
        gs = make(6, segs, first_index=first)
        jobs = cell_jobs(gs, cell, cell, 2048, 3)
        dgs = fr.DeviceGlyphSet(ctx, gs)
        for center in (False, True):
            got, ref, desc = _both(ctx, oracle, gs, jobs, fr.FR_COVERAGE_U8, O.COVERAGE_U8, atlas_shape(len(gs), cell, 3), 2, center, dgs, 0,
                                   ["fr::cov4_kernel<", ", 2> x"])
            assert np.array_equal(got, ref), (segs, cell, center, desc)
            assert set(np.unique(ref)) <= {0x5b, 0, 64, 128, 191, 255}
        dgs.close()


def test_two_by_two_overfull_rows_and_tall_cells(ctx, oracle):
    """n = 2: combs whose rays meet 80 / 12 crossings with 8 / 16 / 32 kept per sample row (the direct sum inside
    cov4_kernel<.., 2>), and a 200 x 1000 cell (2000 sample rows: the 12-bit row fields' limit is 2048)"""
    gl = []
    for teeth in (40, 6):
        cs, box = comb_glyph(teeth)
        gl.append(Glyph(Box(*[int(v) for v in box]), [Contour(c) for c in cs]))
    gs = GlyphSet(gl)
    jobs = cell_jobs(gs, 160, 150, 2048, 2)
    ref = np.full((160, 320), 0x5b, np.uint8)
    oracle.render_batch(gs, jobs, O.COVERAGE_U8, ref, 2, True, 16)
    for kmax in (8, 16, 32):
        try:
            ctx.set_option("kmax", kmax)
            dgs = fr.DeviceGlyphSet(ctx, gs)
            plan = fr.Plan(dgs, jobs, fr.FR_COVERAGE_U8, 2, fr.FR_SAMPLE_CENTER)
            assert plan.stats()["jobs_general"] == 0
            plan.close()
            got = np.full_like(ref, 0x5b)
            rg.render_batch(dgs, jobs, fr.FR_COVERAGE_U8, got, 2, fr.FR_SAMPLE_CENTER)
            dgs.close()
        finally:
            ctx.set_option("kmax", 32)
        assert np.array_equal(got, ref), kmax
    gs = synth_glyphset(1, 96, first_index=4141)
    tall = rg.make_jobs([(0, int(np.floor(gs.boxes[0][0] * 0.1)) - 3, int(np.ceil(gs.boxes[0][3] * 0.5)) + 8, 200, 1000, 3, 1, np.float32(0.5))])
    dgs = fr.DeviceGlyphSet(ctx, gs)
    got, ref, desc = _both(ctx, oracle, gs, tall, fr.FR_COVERAGE_U8, O.COVERAGE_U8, (1003, 210), 2, False, dgs)
    dgs.close()
    assert np.array_equal(got, ref) and (ref == 255).any()


def test_strip_px_option_bounds_the_fast_strips(ctx, oracle, ascii_set):
    """strip_px = 64 / 128: wide cells walk more, narrower strips on the fast kernels; below 64 they take the general kernel"""
    gs = ascii_set.gs
    jobs, H = glyph_dims_jobs(gs, 200, ascii_set.g_upm, 2048, n_glyphs=40)
    ref = np.full((H, 2048), 0x5b, np.uint8)
    oracle.render_batch(gs, jobs, O.COVERAGE_U8, ref, 4, True, 16)
    for px, inst in ((64, "cov4_kernel<2"), (128, "cov4_kernel<3"), (48, "render_kernel")):
        try:
            ctx.set_option("strip_px", px)
            dgs = fr.DeviceGlyphSet(ctx, gs)
            plan = fr.Plan(dgs, jobs, fr.FR_COVERAGE_U8, 4, fr.FR_SAMPLE_CENTER)
            desc = plan.describe()
            plan.close()
            got = np.full_like(ref, 0x5b)
            rg.render_batch(dgs, jobs, fr.FR_COVERAGE_U8, got, 4, fr.FR_SAMPLE_CENTER)
            dgs.close()
        finally:
            ctx.set_option("strip_px", 256)
        assert inst in desc and ("cov4_kernel<4" not in desc), (px, desc)
        assert np.array_equal(got, ref), px


def test_sdf_on_ragged_cells_and_mixed_plans(ctx, oracle, ascii_set):
    """FR_SDF_U8 with the sign travelling as one bit per pixel (win1_kernel's sign-bit mode -> sdf_kernel, which is then the
    only writer of the output): renderGlyph-sized images (ragged, unaligned) of real glyphs, and a plan that mixes fast
    jobs (bit planes) with an 800-segment glyph (general kernel: the sign as a byte in the output), against the CPU twin;
    the bytes around every cell stay untouched"""
    gs = ascii_set.gs
    jobs, H = glyph_dims_jobs(gs, 90, ascii_set.g_upm, 1531, n_glyphs=120)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    for center in (False, True):
        got, ref, desc = _both(ctx, oracle, gs, jobs, fr.FR_SDF_U8, O.SDF_U8, (H + 2, 1531), 1, center, dgs, 0,
                               ["fr::win1_kernel<", ", 3, ", "fr::sdf_kernel"])
        assert np.array_equal(got, ref), (center, desc)
    dgs.close()
    parts = [synth_glyphset(3, 40, first_index=21), synth_glyphset(1, 800, first_index=22), synth_glyphset(2, 100, first_index=23)]
    gs = GlyphSet([p.glyph(i) for p in parts for i in range(len(p))])
    jobs = cell_jobs(gs, 200, 180, 2048, 3)
    jobs["w"] = [200, 131, 64, 200, 33, 190]
    jobs["h"] = [200, 77, 200, 150, 31, 200]
    dgs = fr.DeviceGlyphSet(ctx, gs)
    got, ref, desc = _both(ctx, oracle, gs, jobs, fr.FR_SDF_U8, O.SDF_U8, atlas_shape(len(gs), 200, 3), 1, True, dgs, 1, ["render_kernel", "win1_kernel"])
    dgs.close()
    assert np.array_equal(got, ref), desc


def test_full_size_properties_config5_shard_of_the_real_font(ctx, oracle):
    """BASELINE configs[4] as stated — per-pixel SDF at 512 x 512 per glyph, whole font — at one GPU's share of
    DejaVuSerif-Italic through the C-side producer (the first 384 loadable glyphs): deterministic (two renders equal),
    8 sub-shards == unsharded, culls on == culls off, four cells == the CPU twin."""
    import os
    import torch
    path = "/usr/local/lib/python3.10/dist-packages/matplotlib/mpl-data/fonts/ttf/DejaVuSerif-Italic.ttf"
    if not os.path.exists(path):
        pytest.skip("DejaVuSerif-Italic.ttf not in this image")
    font = fr.Font.initTTF(path)
    gs_all, kept = font.glyphset()
    G, cell, cols = 384, 512, 16
    gs = gs_all.subset(0, G)
    upm = font.information.units_per_em
    jobs = cell_jobs(gs, cell, cell, upm, cols)
    H, W = atlas_shape(G, cell, cols)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    plan = fr.Plan(dgs, jobs, fr.FR_SDF_U8, 1, fr.FR_SAMPLE_CENTER)
    assert plan.pixels == G * cell * cell and plan.stats()["jobs_general"] == 0, plan.describe()
    first = torch.full((H, W), 0x5b, dtype=torch.uint8, device="cuda")
    again = torch.full((H, W), 0xa4, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    plan.render(first.data_ptr(), W, H); plan.render(again.data_ptr(), W, H); ctx.sync()
    assert torch.equal(first, again)                                    # every byte written, the same both times
    plan.close()
    # 8 sub-shards (the 8 GPUs of configs[4]) into the unsharded atlas
    from font_renderer_amd.shard import shard_ranges
    out2 = torch.zeros_like(first)
    for lo, hi in shard_ranges(G, 8):
        sub = gs.subset(lo, hi)
        sd = fr.DeviceGlyphSet(ctx, sub)
        sj = cell_jobs(sub, cell, cell, upm, cols).copy()
        sj["out_x"] = ((np.arange(lo, hi) % cols) * cell).astype(np.uint32)
        sj["out_y"] = ((np.arange(lo, hi) // cols) * cell).astype(np.uint32)
        sp = fr.Plan(sd, sj, fr.FR_SDF_U8, 1, fr.FR_SAMPLE_CENTER)
        sp.render(out2.data_ptr(), W, H); ctx.sync()
        sp.close(); sd.close()
    assert torch.equal(out2, first)
    try:
        ctx.set_option("sdf_cull", 0)
        few = jobs[:32]
        p2 = fr.Plan(dgs, few, fr.FR_SDF_U8, 1, fr.FR_SAMPLE_CENTER)
        out2.zero_(); torch.cuda.synchronize()
        p2.render(out2.data_ptr(), W, H); ctx.sync()
        p2.close()
    finally:
        ctx.set_option("sdf_cull", 1)
    assert torch.equal(out2[:1024], first[:1024])                       # (32 cells = two cell rows)
    host = first.cpu().numpy()
    for k in (5, 37, 200, 383):
        ref = np.zeros((cell, cell), np.uint8)
        one = jobs[k:k + 1].copy()
        one["out_x"] = 0; one["out_y"] = 0
        oracle.render_batch(gs, one, O.SDF_U8, ref, 1, True, 16)
        y, x = int(jobs["out_y"][k]), int(jobs["out_x"][k])
        assert np.array_equal(host[y:y + cell, x:x + cell], ref), k
    dgs.close()
    assert 0.0 < float((first == 0).float().mean()) < 1.0 and 0.0 < float((first == 255).float().mean()) < 1.0


@pytest.mark.parametrize("segs", [400, 512, 700])
def test_glyphs_of_385_to_768_segments_take_the_1024_record_instances(ctx, oracle, segs):
    """glyphs the round-2 library left to the general kernel (more than 384 segments): cov4_kernel / win1_kernel with 16
    records per lane (1024 record slots, two workgroups per CU), ragged cells, every mode == the oracle"""
    gs = GlyphSet([synth_glyphset(2, segs, first_index=3000 + segs).glyph(i) for i in range(2)] + [stroke_glyphset(1, segs, first_index=77).glyph(0)])
    jobs = cell_jobs(gs, 256, 240, 2048, 3)
    jobs["w"] = [256, 200, 77]
    jobs["h"] = [256, 131, 250]
    dgs = fr.DeviceGlyphSet(ctx, gs)
    for mode, omode, n, center in [ONE[1], ONE[0], (fr.FR_COVERAGE_U8, O.COVERAGE_U8, 4, True), (fr.FR_COVERAGE_U8, O.COVERAGE_U8, 2, False), (fr.FR_SDF_U8, O.SDF_U8, 1, True)]:
        got, ref, desc = _both(ctx, oracle, gs, jobs, mode, omode, atlas_shape(len(gs), 256, 3), n, center, dgs, 0, [", 16" if n == 1 else ", 16, "])
        assert np.array_equal(got, ref), (segs, mode, n, desc)
    dgs.close()


def test_whole_font_at_renderglyph_sizes_properties(ctx, oracle):
    """the reference's literal product on a whole real font (every loadable glyph of DejaVuSerif-Italic through the C-side
    producer, three font sizes, ~9 400 images shelf-packed): nothing on the general kernel, two renders equal, the bytes
    around the images untouched, and 90 images spread over the batch == the oracle (gray map and 16-sample coverage)."""
    import os
    import torch
    path = "/usr/local/lib/python3.10/dist-packages/matplotlib/mpl-data/fonts/ttf/DejaVuSerif-Italic.ttf"
    if not os.path.exists(path):
        pytest.skip("DejaVuSerif-Italic.ttf not in this image")
    font = fr.Font.initTTF(path)
    gs, kept = font.glyphset()
    upm = font.information.units_per_em
    parts, y, W = [], 0, 4099
    for size in (16, 40, 100):
        jb, h = glyph_dims_jobs(gs, size, upm, W)
        jb["out_y"] += y
        y += h + 1                                       # (a row of sentinel bytes between the runs of shelves)
        parts.append(jb)
    jobs = np.concatenate(parts)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(len(jobs), 90, replace=False))
    for mode, omode, n, center in ((fr.FR_GRAY_DEBUG, O.GRAY_DEBUG, 1, False), (fr.FR_COVERAGE_U8, O.COVERAGE_U8, 4, True)):
        plan = fr.Plan(dgs, jobs, mode, n, fr.FR_SAMPLE_CENTER if center else fr.FR_SAMPLE_CORNER)
        assert plan.stats()["jobs_general"] == 0, plan.describe()
        a = torch.full((y, W), 0x5b, dtype=torch.uint8, device="cuda")
        b = torch.full((y, W), 0x5b, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        plan.render(a.data_ptr(), W, y); plan.render(b.data_ptr(), W, y); ctx.sync()
        plan.close()
        assert torch.equal(a, b)
        host = a.cpu().numpy()
        covered = np.zeros((y, W), bool)
        for j in jobs:
            covered[j["out_y"]:j["out_y"] + j["h"], j["out_x"]:j["out_x"] + j["w"]] = True
        assert (host[~covered] == 0x5b).all()                                  # nothing outside the images was written
        for k in pick:
            j = jobs[k:k + 1].copy()
            ref = np.zeros((int(j["h"][0]), int(j["w"][0])), np.uint8)
            one = j.copy(); one["out_x"] = 0; one["out_y"] = 0
            oracle.render_batch(gs, one, omode, ref, n, center, 4)
            got = host[j["out_y"][0]:j["out_y"][0] + j["h"][0], j["out_x"][0]:j["out_x"][0] + j["w"][0]]
            assert np.array_equal(got, ref), (mode, int(k))
    dgs.close()


def test_graph_replay_equals_plain_launches(ctx, oracle, ascii_set):
    """option "graph": a plan's launches (several classes, with and without the fork onto the second stream, the SDF's
    three kernels) captured into a hipGraph at the first render to a destination and replayed afterwards — every replay,
    a second destination (re-capture) and a render after an option change == the plain launches, byte for byte."""
    import torch
    gs = ascii_set.gs
    W = 1021
    parts, y = [], 0
    for size in (20, 64, 150):
        jb, h = glyph_dims_jobs(gs, size, ascii_set.g_upm, W)
        jb["out_y"] += y
        y += h + 1
        parts.append(jb)
    jobs = np.concatenate(parts)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    try:
        for mode, n, center, overlap in ((fr.FR_GRAY_DEBUG, 1, False, 1), (fr.FR_COVERAGE_U8, 4, True, 2), (fr.FR_COVERAGE_U8, 2, True, 1),
                                         (fr.FR_SDF_U8, 1, True, 2), (fr.FR_WINDING_I16, 1, False, 2)):
            dt = torch.int16 if mode == fr.FR_WINDING_I16 else torch.uint8
            ctx.set_option("overlap", overlap)
            plan = fr.Plan(dgs, jobs, mode, n, fr.FR_SAMPLE_CENTER if center else fr.FR_SAMPLE_CORNER)
            assert plan.describe().count("fr::") >= 2, plan.describe()          # several launches per render
            want = torch.full((y, W), 0x5b, dtype=dt, device="cuda")
            ctx.set_option("graph", 0)
            plan.render(want.data_ptr(), W, y); ctx.sync()
            ctx.set_option("graph", 1)
            a = torch.full((y, W), 0x5b, dtype=dt, device="cuda")
            b = torch.full((y, W), 0x5b, dtype=dt, device="cuda")
            torch.cuda.synchronize()
            plan.render(a.data_ptr(), W, y); ctx.sync()                          # capture + first launch
            assert torch.equal(a, want), mode
            a.fill_(0x5b); torch.cuda.synchronize()
            plan.render(a.data_ptr(), W, y); plan.render(a.data_ptr(), W, y); ctx.sync()     # replays
            assert torch.equal(a, want), mode
            plan.render(b.data_ptr(), W, y); ctx.sync()                          # another destination: captured again
            assert torch.equal(b, want), mode
            ctx.set_option("overlap", 0)                                         # an option changed: captured again
            a.fill_(0x5b); torch.cuda.synchronize()
            ms = plan.render_timed(a.data_ptr(), W, y)
            assert ms > 0 and torch.equal(a, want), mode
            plan.close()
    finally:
        ctx.set_option("graph", 0)
        ctx.set_option("overlap", 1)
        dgs.close()


def _comb_with_diamonds(teeth, n_diamonds, x_lo, x_hi, y_lo, y_hi, seed):
    """a comb (2 * teeth crossings per ray through the teeth) + small diamond contours that only add SEGMENTS (and so move
    the glyph into the fast kernels' 512-record instances) — spread over [x_lo, x_hi] x [y_lo, y_hi]"""
    cs, _ = comb_glyph(teeth)
    rng = np.random.default_rng(seed)
    contours = [Contour(c) for c in cs]
    for _ in range(n_diamonds):
        cx, cy, r = int(rng.integers(x_lo, x_hi)), int(rng.integers(y_lo, y_hi)), int(rng.integers(6, 30))
        poly = np.array([(cx - r, cy), (cx, cy + r), (cx + r, cy), (cx, cy - r)], np.int64)
        nxt = np.roll(poly, -1, 0)
        pts = np.empty((9, 2), np.int64)
        pts[0:-1:2] = poly
        pts[1:-1:2] = (poly + nxt) // 2
        pts[-1] = poly[0]
        contours.append(Contour(pts.astype(np.int16)))
    allp = np.concatenate([c.points for c in contours])
    return Glyph(Box(int(allp[:, 0].min()), int(allp[:, 1].min()), int(allp[:, 0].max()), int(allp[:, 1].max())), contours)


def test_bands_of_overfull_rows_walk_in_two_halves(ctx, oracle):
    """the 512-record instances (glyphs of 257 .. 384 segments) walk a band with three or more over-full rows twice more —
    right half of the sample columns, then left half, 32 slots each — instead of one direct sum per row.  Glyphs whose
    rays meet 50 crossings spread over the whole width (both halves fit), the same comb squeezed into the left 45 % of
    the cell (the left half overflows: those rows still take the direct sum), 70 crossings, and wiggly synthetic
    outlines of 300 segments; 4 x 4 and 2 x 2 samples, 256- / 128- / 64-pixel strips, ragged cells — all == the oracle."""
    gl = [_comb_with_diamonds(25, 45, 150, 1850, 110, 290, 1),          # 50 crossings per ray, 25 + 25
          _comb_with_diamonds(25, 45, 2000, 4100, 100, 1600, 2),        # the comb in the left half: 50 + 2
          _comb_with_diamonds(35, 30, 150, 1850, 110, 290, 3),          # 70 crossings per ray: 35 + 35, both halves overflow
          _comb_with_diamonds(20, 50, 150, 1850, 110, 1600, 4)]         # 40 (+ a few) crossings
    sg = synth_glyphset(4, 300, first_index=4242)
    gl += [sg.glyph(i) for i in range(len(sg))]
    gs = GlyphSet(gl)
    # every glyph fills its cell's width (the cell's origin is the glyph's own lower-left corner)
    upm = np.array([int((g.box.x_max - g.box.x_min) * 1.04) for g in gl[:4]] + [2048] * len(sg), np.uint16)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    for cell_w, cell_h, cols in ((256, 256, 4), (128, 128, 4), (250, 141, 3), (61, 77, 5)):
        cell = max(cell_w, cell_h)
        jobs = cell_jobs(gs, cell, cell_w, upm, cols)
        jobs["w"] = cell_w; jobs["h"] = cell_h
        shape = (int((jobs["out_y"] + jobs["h"]).max()) + 1, int((jobs["out_x"] + jobs["w"]).max()) + 3)
        for n in (4, 2):
            got, ref, desc = _both(ctx, oracle, gs, jobs, fr.FR_COVERAGE_U8, O.COVERAGE_U8, shape, n, True, dgs, 0, ["cov4_kernel<", ", 32, 8, "])
            assert np.array_equal(got, ref), (cell_w, cell_h, n, desc)
    dgs.close()


def test_row_pitch_limit_is_checked(ctx, ascii_set):
    """the fast kernels address the rows of a wave band by 32-bit offsets from the band's base: fr_plan_render refuses a
    pitch above 2^26 elements (FR_E_INVALID) instead of rendering somewhere else"""
    import torch
    gs = ascii_set.gs
    jobs, H = glyph_dims_jobs(gs, 20, ascii_set.g_upm, 509, n_glyphs=4)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    plan = fr.Plan(dgs, jobs, fr.FR_GRAY_DEBUG, 1, fr.FR_SAMPLE_CORNER)
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception) as e:
        plan.render(buf.data_ptr(), (1 << 26) + 1, H)
    assert "pitch" in str(e.value)
    plan.close()
    dgs.close()
