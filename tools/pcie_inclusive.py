#!/usr/bin/env python3
"""The boundary's one-shot HOST form, timed end to end: fr_render_batch(host buffer) = hipMalloc + H2D of the caller's
buffer (pixels outside the jobs keep its bytes) + the plan + the render + D2H + free.  Never bench.py's `value` (that one
starts with everything resident in HBM); DESIGN.md section 7 quotes this number beside it.
usage: python tools/pcie_inclusive.py [workload] [glyphs]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import bench  # noqa: E402
import font_renderer_amd as fr  # noqa: E402
from font_renderer_amd import render_glyph as rg  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3_cjk21k_256px_s128_16spp"
wl = dict(bench.WORKLOADS[name])
G = int(sys.argv[2]) if len(sys.argv) > 2 else wl["glyphs"]
gs, jobs, (H, W) = bench.build_inputs(wl, 0, 0, G)
ctx = fr.Context(0)
dgs = fr.DeviceGlyphSet(ctx, gs)
out = np.zeros((H, W), np.uint8)
mode, n = fr.FR_COVERAGE_U8, 4
rg.render_batch(dgs, jobs, mode, out, n, fr.FR_SAMPLE_CENTER)        # (first call: pages in the buffer, warms the clocks)
ts = []
for _ in range(5):
    t = time.perf_counter()
    rg.render_batch(dgs, jobs, mode, out, n, fr.FR_SAMPLE_CENTER)
    ts.append(time.perf_counter() - t)
px = int((jobs["w"].astype(np.int64) * jobs["h"]).sum())
best = min(ts)
print(json.dumps({"workload": name, "jobs": len(jobs), "pixels": px, "host_buffer_bytes": int(out.nbytes), "memory": "pageable numpy array",
                  "ms_best_of_5": round(best * 1e3, 2), "ms_all": [round(t * 1e3, 2) for t in ts],
                  "Gpixel_per_s_pcie_inclusive": round(px / best / 1e9, 2),
                  "GBps_over_the_bus": round(2 * out.nbytes / best / 1e9, 2)}))
