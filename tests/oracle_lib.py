"""ctypes wrapper of oracle/_build/libfr_oracle.so — the CHECKER for the tests.
Never imported by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ODIR, "_build", "libfr_oracle.so")

WINDING_I16, GRAY_DEBUG, MASK_NONZERO, COVERAGE_U8, SDF_U8 = 0, 1, 2, 3, 4


def build():
    src = [os.path.join(ODIR, f) for f in ("fr_oracle.c", "fr_oracle.h", "Makefile")]
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in src):
        subprocess.check_call(["make", "-C", ODIR, "-s"])
    return SO


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self):
        self.lib = L = C.CDLL(build())
        V = C.c_void_p
        L.or_glyph_winding_at.restype = C.c_int16
        L.or_glyph_winding_at.argtypes = [V, V, C.c_uint32, C.c_float, C.c_float]
        L.or_render_glyph_dims.argtypes = [V, C.c_uint16, C.c_uint16, V, V, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.POINTER(C.c_float)]
        L.or_render_glyph.argtypes = [V, V, C.c_uint32, V, C.c_uint16, C.c_uint16, V]
        L.or_render_cell.restype = C.c_int
        L.or_render_cell.argtypes = [V, V, C.c_uint32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_int, C.c_int, V, C.c_size_t]
        L.or_render_batch.restype = C.c_int
        L.or_render_batch.argtypes = [V, V, V, V, C.c_uint32, C.c_int, C.c_int, C.c_int, V, C.c_size_t, C.c_int]
        L.or_count_ttf_points.restype = C.c_uint32
        L.or_count_ttf_points.argtypes = [V, C.c_uint32, V]
        L.or_contour_init_ttf.restype = C.c_uint32
        L.or_contour_init_ttf.argtypes = [V, V, V, C.c_uint32, V]
        L.or_transform_point.restype = C.c_int
        L.or_transform_point.argtypes = [C.c_int16, C.c_int16, V, C.c_int16, C.c_int16, C.c_int, V, V]
        L.or_glyph_info_init.argtypes = [V, V, C.c_uint32, V, V]
        L.or_winding_in_glyph.restype = C.c_int16
        L.or_winding_in_glyph.argtypes = [V, V, C.c_uint32, V, V, C.c_int16, C.c_int16]
        L.or_winding_lattice.argtypes = [V, V, C.c_uint32, V, V]
        L.or_exact_lattice.argtypes = [V, V, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, V]
        L.or_exact_lattice.restype = None
        L.or_exact_coverage.argtypes = [V, V, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, V]
        L.or_exact_coverage.restype = None
        L.or_glyph_debug_render.argtypes = [V, V, C.c_uint32, V, C.c_uint8, V]
        L.or_winding_rgb.argtypes = [C.c_int16, C.c_uint8, C.c_uint8, V]
        L.or_qoi_encode_rgb.restype = C.c_size_t
        L.or_qoi_encode_rgb.argtypes = [V, C.c_uint32, C.c_uint32, V, C.c_size_t]
        L.or_diag_i64_overflow_count.restype = C.c_uint64
        L.or_diag_assert_fail_count.restype = C.c_uint64

    # -- glyph helpers: pts (n,2) i16, cstart (c+1,) u32
    @staticmethod
    def _g(glyph):
        from font_renderer_amd.glyph import GlyphSet
        gs = GlyphSet([glyph])
        return gs.points_xy, gs.contour_start, gs.n_contours

    def winding_at(self, glyph, cx, cy) -> int:
        pts, cs, nc = self._g(glyph)
        return int(self.lib.or_glyph_winding_at(_p(pts), _p(cs), nc, np.float32(cx), np.float32(cy)))

    def render_glyph_dims(self, box, upm, font_size):
        b = np.asarray(box, np.int16)
        mn, mx = np.zeros(2, np.int16), np.zeros(2, np.int16)
        w, h, s = C.c_uint16(), C.c_uint16(), C.c_float()
        self.lib.or_render_glyph_dims(_p(b), upm, font_size, _p(mn), _p(mx), C.byref(w), C.byref(h), C.byref(s))
        return (int(mn[0]), int(mn[1])), (int(mx[0]), int(mx[1])), w.value, h.value, s.value

    def render_glyph(self, glyph, upm, font_size) -> np.ndarray:
        pts, cs, nc = self._g(glyph)
        box = glyph.box.as_array()
        _, _, w, h, _ = self.render_glyph_dims(box, upm, font_size)
        out = np.zeros((h, w), np.uint8)
        self.lib.or_render_glyph(_p(pts), _p(cs), nc, _p(box), upm, font_size, _p(out))
        return out

    def render_cell(self, glyph, min_x, max_y, w, h, scale, mode, n=1, center=False) -> np.ndarray:
        pts, cs, nc = self._g(glyph)
        out = np.zeros((h, w), np.int16 if mode == WINDING_I16 else np.uint8)
        rc = self.lib.or_render_cell(_p(pts), _p(cs), nc, min_x, max_y, w, h, np.float32(scale), mode, n, int(center), _p(out), w)
        assert rc == 0
        return out

    def render_batch(self, gs, jobs, mode, out, n=1, center=False, threads=1) -> np.ndarray:
        """gs: GlyphSet, jobs: fr_job-layout array (same 32-byte layout as or_job)"""
        jobs = np.ascontiguousarray(jobs)
        assert jobs.dtype.itemsize == 32 and out.flags.c_contiguous
        rc = self.lib.or_render_batch(_p(gs.points_xy), _p(gs.contour_start), _p(gs.glyph_start), _p(jobs), len(jobs),
                                      mode, n, int(center), _p(out), out.shape[1], threads)
        assert rc == 0
        return out

    def expand_contours(self, coords, on, ends):
        coords = np.ascontiguousarray(coords, np.int16); on = np.ascontiguousarray(on, np.uint8)
        ends = np.ascontiguousarray(ends, np.uint16)
        total = self.lib.or_count_ttf_points(_p(ends), len(ends), _p(on))
        out, res = np.zeros((total + 2, 2), np.int16), []
        o = 0
        for ci in range(len(ends)):
            n = self.lib.or_contour_init_ttf(_p(coords), _p(on), _p(ends), ci, _p(out[o:]))
            res.append(out[o:o + n].copy())
            o += n
        assert o == total
        return res

    def transform_point(self, x, y, m, e, f, round_to_grid):
        m = np.asarray(m, np.int16); ox, oy = C.c_int16(), C.c_int16()
        rc = self.lib.or_transform_point(x, y, _p(m), e, f, int(round_to_grid), C.byref(ox), C.byref(oy))
        return rc, ox.value, oy.value

    def glyph_info(self, glyph):
        pts, cs, nc = self._g(glyph)
        n = glyph.curve_count
        ct, ip = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
        self.lib.or_glyph_info_init(_p(pts), _p(cs), nc, _p(ct), _p(ip))
        return ct[:n], ip[:n]

    def winding_in_glyph(self, glyph, queries) -> np.ndarray:
        pts, cs, nc = self._g(glyph)
        ct, ip = self.glyph_info(glyph)
        ct = np.ascontiguousarray(ct) if len(ct) else np.zeros(1, np.uint8)
        ip = np.ascontiguousarray(ip) if len(ip) else np.zeros(1, np.uint8)
        q = np.asarray(queries, np.int16).reshape(-1, 2)
        return np.array([self.lib.or_winding_in_glyph(_p(pts), _p(cs), nc, _p(ct), _p(ip), int(x), int(y)) for x, y in q], np.int16)

    def winding_lattice(self, glyph) -> np.ndarray:
        pts, cs, nc = self._g(glyph)
        box = glyph.box.as_array()
        W, H = int(box[2]) - int(box[0]) + 3, int(box[3]) - int(box[1]) + 3
        out = np.zeros((H, W), np.int16)
        self.lib.or_winding_lattice(_p(pts), _p(cs), nc, _p(box), _p(out))
        return out

    def exact_lattice(self, glyph, K, x0, y0, w, h) -> np.ndarray:
        pts, cs, nc = self._g(glyph)
        out = np.zeros((h, w), np.int16)
        self.lib.or_exact_lattice(_p(pts), _p(cs), nc, K, x0, y0, w, h, _p(out))
        return out

    def exact_coverage(self, glyph, K, x0, y0, w_px, h_px, n) -> np.ndarray:
        pts, cs, nc = self._g(glyph)
        out = np.zeros((h_px, w_px), np.uint8)
        self.lib.or_exact_coverage(_p(pts), _p(cs), nc, K, x0, y0, w_px, h_px, n, _p(out))
        return out

    def glyph_debug_render(self, glyph, winding_scale) -> np.ndarray:
        pts, cs, nc = self._g(glyph)
        box = glyph.box.as_array()
        W, H = int(box[2]) - int(box[0]) + 3, int(box[3]) - int(box[1]) + 3
        out = np.zeros((H, W, 3), np.uint8)
        self.lib.or_glyph_debug_render(_p(pts), _p(cs), nc, _p(box), winding_scale, _p(out))
        return out

    def winding_rgb(self, val, scaler, overflow):
        rgb = np.zeros(3, np.uint8)
        self.lib.or_winding_rgb(val, scaler, overflow, _p(rgb))
        return tuple(int(v) for v in rgb)

    def qoi_encode(self, rgb: np.ndarray) -> bytes:
        rgb = np.ascontiguousarray(rgb, np.uint8)
        h, w = rgb.shape[:2]
        cap = 14 + 8 + 4 * w * h + 16
        out = np.zeros(cap, np.uint8)
        n = self.lib.or_qoi_encode_rgb(_p(rgb), w, h, _p(out), cap)
        assert n > 0
        return out[:n].tobytes()

    def diag(self):
        return int(self.lib.or_diag_i64_overflow_count()), int(self.lib.or_diag_assert_fail_count())

    def diag_reset(self):
        self.lib.or_diag_reset()
