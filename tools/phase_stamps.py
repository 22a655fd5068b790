#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of render_kernel (wave 0 of every workgroup),
from the STAMPS build (make -C font-renderer_amd/csrc stamps).  Shares only — the stamped
build's run time is never quoted."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import font_renderer_amd as fr
from font_renderer_amd import _lib
_lib.lib_path = lambda: os.path.join(ROOT, "font-renderer_amd", "libfr_raster_stamps.so")
import torch
from font_renderer_amd.atlas import atlas_shape, cell_jobs
from font_renderer_amd.synth import synth_glyphset
G, cell, S, n, cols = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 256, 128, 4, 64
lib = fr.load_library()
gs = synth_glyphset(G, S)
ctx = fr.Context(0)
dgs = fr.DeviceGlyphSet(ctx, gs)
H, W = atlas_shape(G, cell, cols)
out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
plan = fr.Plan(dgs, cell_jobs(gs, cell, cell, 2048, cols), fr.FR_COVERAGE_U8, n, fr.FR_SAMPLE_CENTER)
buf = (C.c_ulonglong * 16)()
plan.render(out.data_ptr(), W, H); ctx.sync()
lib.fr_debug_read_stamps(buf, 1)
plan.render(out.data_ptr(), W, H); ctx.sync()
lib.fr_debug_read_stamps(buf, 1)
v = np.array(list(buf), float)
names = ["setup (records, ranges, cx table)", "pair layout", "pair evaluation", "list pull + sort", "zero + toggles", "windows + stores"]
v6 = v[:6]
for nme, x in zip(names, v6):
    print(f"{nme:36s} {x / v6.sum() * 100:6.2f} %   {x / G:10.0f} cycles/workgroup (wave 0)")
bands = max(v[8], 1)
print(f"wave bands {v[8]:.0f}; per wave band: pairs {v[9] / bands:.1f}, crossings ~{64 * v[11] / bands:.1f}")
t = int(v[10])
print("sort tier per wave band (lane 0's wave bands x 2^16 packing): <=8:", t & 0xffff, " 9-12:", (t >> 16) & 0xffff, " 13-16:", (t >> 32) & 0xffff, " >16:", (t >> 48) & 0xffff)
