#!/usr/bin/env python3
"""configs[0] latency of ONE library build, ABI-level: fr_render_glyph on STIX 'A' at font_size 64 (47 x 45).
usage: tools/c1_latency.py <path to a libfr_raster.so>  (any round's build: binds only fr_ctx_create /
fr_render_glyph / fr_ctx_destroy)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import fixtures  # noqa: E402
from font_renderer_amd.glyph import GlyphSet  # noqa: E402

try:
    import torch  # noqa: F401  (one HIP runtime per process: torch's first, as font_renderer_amd._lib does)
    C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=C.RTLD_GLOBAL)
except Exception:
    pass
lib = C.CDLL(sys.argv[1])
asc = fixtures.load_ascii()
g = asc.glyph(asc.find("STIX", "A"))
gs = GlyphSet([g])
pts, cs, box = gs.points_xy, gs.contour_start, g.box.as_array()
P = lambda a: a.ctypes.data_as(C.c_void_p)
ctx = C.c_void_p()
assert lib.fr_ctx_create(0, None, C.byref(ctx)) == 0
buf = np.zeros(47 * 45, np.uint8)
lib.fr_render_glyph.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint16, C.c_uint16, C.c_int32, C.c_void_p]
call = lambda: lib.fr_render_glyph(ctx, P(pts), P(cs), len(cs) - 1, P(box), 1000, 64, 1, P(buf))
for _ in range(50):
    assert call() == 0
t = time.perf_counter()
for _ in range(500):
    call()
cold = (time.perf_counter() - t) / 500 * 1e6      # the first few hundred calls run while the GPU clocks ramp up
for _ in range(1500):
    call()
t = time.perf_counter()
for _ in range(1000):
    call()
us = (time.perf_counter() - t) / 1000 * 1e6
import hashlib
print(f"{os.path.basename(sys.argv[1])}: {us:.1f} us per fr_render_glyph call in steady state (first 500 calls: {cold:.1f} us) — STIX 'A' at 64 -> 47x45; "
      f"sha256[:16] of the image {hashlib.sha256(buf.tobytes()).hexdigest()[:16]}")
if hasattr(lib, "fr_ctx_set_option") and len(sys.argv) > 2:
    lib.fr_ctx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    if lib.fr_ctx_set_option(ctx, b"zero_copy", 1) == 0:
        for _ in range(200):
            call()
        t = time.perf_counter()
        for _ in range(1000):
            call()
        print(f"   with zero_copy = 1 (kernel reads / writes pinned host memory): {(time.perf_counter() - t) / 1000 * 1e6:.1f} us per call")
lib.fr_ctx_destroy(ctx)
