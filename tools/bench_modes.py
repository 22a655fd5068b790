#!/usr/bin/env python3
"""Throughput of the 1-sample modes (the reference's own renderGlyph value map FR_GRAY_DEBUG, winding
values, non-zero mask) on C3-shaped cells — not the headline metric; run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import font_renderer_amd as fr
from font_renderer_amd.atlas import atlas_shape, cell_jobs
from font_renderer_amd.synth import synth_glyphset
G, cell, S, cols = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 256, 128, 64
gs = synth_glyphset(G, S)
stream = torch.cuda.Stream()
ctx = fr.Context(0, stream.cuda_stream)
dgs = fr.DeviceGlyphSet(ctx, gs)
H, W = atlas_shape(G, cell, cols)
jobs = cell_jobs(gs, cell, cell, 2048, cols)
for name, mode, dt in (("GRAY_DEBUG", fr.FR_GRAY_DEBUG, torch.uint8), ("WINDING_I16", fr.FR_WINDING_I16, torch.int16),
                       ("MASK_NONZERO", fr.FR_MASK_NONZERO, torch.uint8), ("COVERAGE_U8 n=1", fr.FR_COVERAGE_U8, torch.uint8)):
    with torch.cuda.stream(stream):
        out = torch.zeros((H, W), dtype=dt, device="cuda")
    plan = fr.Plan(dgs, jobs, mode, 1, fr.FR_SAMPLE_CORNER)
    ms = sorted(plan.render_timed(out.data_ptr(), W, H) for _ in range(10))
    print(f"{name:16s} {np.mean(ms[:8]):8.4f} ms  {plan.pixels / np.mean(ms[:8]) / 1e6:9.1f} Gpixel/s")
    plan.close()
