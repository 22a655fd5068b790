#!/usr/bin/env python3
"""Sort-tier histogram and over-full-row share of cov4_kernel on a bench workload (diagnostic library
libfr_raster_var_c4stats.so: make -C font-renderer_amd/csrc variant NAME=c4stats DEFS=-DFR_C4_STATS).
usage: FR_RASTER_LIB=font-renderer_amd/libfr_raster_var_c4stats.so python tools/c4_stats.py <workload> [glyphs]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import font_renderer_amd as fr  # noqa: E402

wl = dict(bench.WORKLOADS[sys.argv[1]])
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
gs, jobs, (H, W) = bench.build_inputs(wl, 0, 0, G)
ctx = fr.Context(0)
out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
dgs = fr.DeviceGlyphSet(ctx, gs)
plan = fr.Plan(dgs, jobs, fr.FR_COVERAGE_U8, 4, fr.FR_SAMPLE_CENTER)
lib = fr.load_library()
st = (C.c_ulonglong * 16)()
assert lib.fr_debug_read_c4_stats(st, 1) == 0
plan.render(out.data_ptr(), W, H)
ctx.sync()
assert lib.fr_debug_read_c4_stats(st, 0) == 0
bands, t8, t16, t32, ovf, cross, r8, r16, r32 = [int(st[i]) for i in range(9)]
rows = bands * 64
print(json.dumps({"workload": sys.argv[1], "cells": len(jobs), "wave_bands_with_crossings": bands,
                  "sort_tier_share": {"8_slots": round(t8 / bands, 4), "16_slots": round(t16 / bands, 4), "32_slots": round(t32 / bands, 4)},
                  "crossings_per_sample_row": round(cross / rows, 3),
                  "rows_share": {"gt8": round(r8 / rows, 5), "gt16": round(r16 / rows, 5), "over_full_gt32": round(r32 / rows, 6)},
                  "over_full_rows": ovf,
                  "first_walk": {"bands_with_over_full_rows": int(st[9]), "their_rows": int(st[10]), "bands_walked_in_halves": int(st[11]), "their_rows_": int(st[12])}}))
