#!/usr/bin/env python3
"""Throughput of FR_SDF_U8 on BASELINE configs[4]-shaped cells (512 x 512 per glyph, S = 64) — run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import font_renderer_amd as fr
from font_renderer_amd.atlas import atlas_shape, cell_jobs
from font_renderer_amd.synth import synth_glyphset
G, cell, S, cols = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 512, 64, 16
gs = synth_glyphset(G, S)
stream = torch.cuda.Stream()
ctx = fr.Context(0, stream.cuda_stream)
dgs = fr.DeviceGlyphSet(ctx, gs)
H, W = atlas_shape(G, cell, cols)
jobs = cell_jobs(gs, cell, cell, 2048, cols)
with torch.cuda.stream(stream):
    out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
plan = fr.Plan(dgs, jobs, fr.FR_SDF_U8, 1, fr.FR_SAMPLE_CENTER)
ms = sorted(plan.render_timed(out.data_ptr(), W, H) for _ in range(5))
print(f"SDF {G} glyphs x {cell}^2, S={S}: {np.mean(ms[:4]):8.3f} ms  {plan.pixels / np.mean(ms[:4]) / 1e6:9.2f} Gpixel/s")
