#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity fuzz (not part of the suite): random synthetic and real
glyphs, cell sizes, scales, origins, sample counts / phases, all modes.  Prints mismatches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import font_renderer_amd as fr
import oracle_lib as O
import fixtures
from font_renderer_amd import render_glyph as rg
from font_renderer_amd.glyph import GlyphSet
from font_renderer_amd.synth import stroke_glyphset, synth_glyphset

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(seed)
orc, asc, ctx = O.Oracle(), fixtures.load_ascii(), fr.Context(0)
bad = 0
for case in range(cases):
    kind = rng.random()
    if kind < 0.4:
        S = int(rng.choice([3, 5, 8, 16, 31, 64, 100, 128, 140, 200, 256, 300, 390, 512, 700, 800]))
        gs = synth_glyphset(int(rng.integers(1, 5)), S, first_index=int(rng.integers(0, 1 << 20)))
        upm = 2048
    elif kind < 0.55:
        gs = stroke_glyphset(int(rng.integers(1, 4)), int(rng.choice([48, 96, 160, 256])), first_index=int(rng.integers(0, 1 << 20)))
        upm = 2048
    else:
        i = int(rng.integers(0, len(asc) - 4))
        gs = asc.gs.subset(i, i + int(rng.integers(1, 4)))
        upm = int(asc.g_upm[i])
    mode = int(rng.choice([0, 1, 2, 3, 3, 3, 4]))
    n = int(rng.choice([1, 2, 4])) if mode == 3 else 1
    center = bool(rng.integers(0, 2))
    rows, x = [], int(rng.integers(0, 20))
    uniform = (mode != 4 and n in (1, 4) and rng.random() < 0.6)   # atlas cells: cov4_kernel / win1_kernel (and mixed plans)
    ucell = int(rng.choice([128, 256]))
    if uniform and rng.random() < 0.5:
        x = 16 * int(rng.integers(0, 3))
    for g in range(len(gs)):
        w, h = int(rng.integers(1, 300)), int(rng.integers(1, 200))
        if uniform:
            w, h = ucell * int(rng.integers(1, 3)), 16 * int(rng.integers(1, 20))
            if rng.random() < 0.1:
                w -= int(rng.integers(1, 9))            # one odd cell: the plan becomes mixed
        if mode == 4:
            w, h = min(w, 150), min(h, 90)
        fs = int(rng.integers(4, 400))
        s = np.float32(fs) / np.float32(upm)
        box = gs.boxes[g].astype(np.float32)
        rows.append((g, int(np.floor(box[0] * s)) + int(rng.integers(-6, 6)), int(np.ceil(box[3] * s)) + int(rng.integers(-6, 6)),
                     w, h, x, int(rng.integers(0, 5)), s))
        x += w + (16 * int(rng.integers(0, 2)) if (uniform and rng.random() < 0.5) else int(rng.integers(0, 18)))
    jobs = rg.make_jobs(rows)
    shape = (max(r[4] + r[6] for r in rows) + 2, x + 3)
    dt = np.int16 if mode == 0 else np.uint8
    got, ref = np.full(shape, 9, dt), np.full(shape, 9, dt)
    dgs = fr.DeviceGlyphSet(ctx, gs)
    rg.render_batch(dgs, jobs, mode, got, n, fr.FR_SAMPLE_CENTER if center else fr.FR_SAMPLE_CORNER)
    dgs.close()
    orc.render_batch(gs, jobs, mode, ref, n, center, 16)
    if not np.array_equal(got, ref):
        bad += 1
        d = np.argwhere(got != ref)
        print(f"MISMATCH case {case}: mode {mode} n {n} center {center} glyphs {len(gs)} first diff at {d[0].tolist()} ({len(d)} px)")
print(f"fuzz seed {seed}: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
