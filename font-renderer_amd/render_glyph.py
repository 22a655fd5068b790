"""Host-side mirror of /root/reference/src/tools/render_glyph.zig on the C ABI.

Same names and argument meaning as the reference: renderGlyph(glyph, font_info,
font_size) -> Image.Gray (:11), GlyphInfo.init (:110), windingInGlyph (:160).  The
arithmetic runs in libfr_raster.so (HIP, gfx950); nothing is computed in Python."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib as L
from .glyph import FontInformation, Glyph, GlyphSet
from .image import RGB, Gray, Winding


def _flat(glyph: Glyph):
    gs = GlyphSet([glyph])
    return gs.points_xy, gs.contour_start, len(gs.contour_start) - 1


class Context:
    """fr_ctx.  stream: a hipStream_t handle (int) to launch on, e.g.
    torch.cuda.current_stream().cuda_stream, or None for a private stream."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._lib = L.load_library()
        h = C.c_void_p()
        L.check(self._lib.fr_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h
        self.device = device

    def set_option(self, key: str, value: int) -> None:
        L.check(self._lib.fr_ctx_set_option(self._h, key.encode(), value))

    def sync(self) -> None:
        L.check(self._lib.fr_ctx_sync(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.fr_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class DeviceGlyphSet:
    """fr_glyphset: points in HBM + root records (precompute kernel run at creation)."""

    def __init__(self, ctx: Context, gs: GlyphSet):
        self.ctx, self.host = ctx, gs
        h = C.c_void_p()
        L.check(ctx._lib.fr_glyphset_create(ctx._h, L.ptr(gs.points_xy), L.ptr(gs.contour_start), gs.n_contours,
                                            L.ptr(gs.glyph_start), len(gs), C.byref(h)))
        self._h = h

    def prepare(self) -> None:
        L.check(self.ctx._lib.fr_glyphset_prepare(self._h))

    def stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        L.check(self.ctx._lib.fr_glyphset_stats(self._h, C.byref(a), C.byref(b)))
        return {"segments": a.value, "records": b.value}

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.ctx._lib.fr_glyphset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_jobs(rows: Sequence) -> np.ndarray:
    """rows of (glyph, min_x, max_y, w, h, out_x, out_y, scale) -> fr_job array"""
    dt = np.dtype([("glyph", "<u4"), ("min_x", "<i4"), ("max_y", "<i4"), ("w", "<u4"), ("h", "<u4"),
                   ("out_x", "<u4"), ("out_y", "<u4"), ("scale", "<f4")])
    a = np.zeros(len(rows), dt)
    for i, r in enumerate(rows):
        a[i] = tuple(r)
    return a


class Plan:
    """fr_plan: a job table resident on the device."""

    def __init__(self, dgs: DeviceGlyphSet, jobs: np.ndarray, mode: int, samples_per_axis: int = 1,
                 sample_phase: int = L.FR_SAMPLE_CORNER):
        assert jobs.dtype.itemsize == 32
        self.ctx, self.dgs, self.mode = dgs.ctx, dgs, mode
        self.params = L.RasterParams(mode, samples_per_axis, sample_phase, 0)
        jobs = np.ascontiguousarray(jobs)
        h = C.c_void_p()
        L.check(self.ctx._lib.fr_plan_create(self.ctx._h, dgs._h, L.ptr(jobs), len(jobs), C.byref(self.params), C.byref(h)))
        self._h = h
        self.n_jobs = len(jobs)

    @property
    def pixels(self) -> int:
        return int(self.ctx._lib.fr_plan_pixels(self._h))

    def stats(self) -> dict:
        """how the jobs are split between the two render kernels (decided per job)"""
        a, b = C.c_uint32(), C.c_uint32()
        L.check(self.ctx._lib.fr_plan_stats(self._h, C.byref(a), C.byref(b)))
        return {"jobs_cov4": a.value, "jobs_general": b.value}

    def describe(self) -> str:
        """the kernel instances one render launches, as rocprofv3 names them, with job counts"""
        buf = C.create_string_buffer(1024)
        L.check(self.ctx._lib.fr_plan_describe(self._h, buf, len(buf)))
        return buf.value.decode()

    def render(self, out_dev_ptr: int, out_stride: int, out_rows: int) -> None:
        """asynchronous on the context's stream; out_dev_ptr is a DEVICE address"""
        L.check(self.ctx._lib.fr_plan_render(self._h, C.c_void_p(out_dev_ptr), out_stride, out_rows))

    def render_timed(self, out_dev_ptr: int, out_stride: int, out_rows: int) -> float:
        ms = C.c_float()
        L.check(self.ctx._lib.fr_plan_render_timed(self._h, C.c_void_p(out_dev_ptr), out_stride, out_rows, C.byref(ms)))
        return ms.value

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.ctx._lib.fr_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_batch(dgs: DeviceGlyphSet, jobs: np.ndarray, mode: int, out: np.ndarray, samples_per_axis: int = 1,
                 sample_phase: int = L.FR_SAMPLE_CORNER) -> np.ndarray:
    """fr_render_batch into a HOST array `out` (2-D, u8 or i16 for FR_WINDING_I16)."""
    want = np.int16 if mode == L.FR_WINDING_I16 else np.uint8
    assert out.dtype == want and out.ndim == 2 and out.flags.c_contiguous
    prm = L.RasterParams(mode, samples_per_axis, sample_phase, 0)
    jobs = np.ascontiguousarray(jobs)
    L.check(dgs.ctx._lib.fr_render_batch(dgs.ctx._h, dgs._h, L.ptr(jobs), len(jobs), C.byref(prm), L.ptr(out),
                                         out.shape[1], out.shape[0]))
    return out


def render_glyph_dims(box, units_per_em: int, font_size: int):
    """render_glyph.zig:13-19 -> (min_corner, max_corner, width, height, scale)"""
    lib = L.load_library()
    b = np.asarray(box, np.int16)
    mn, mx = np.zeros(2, np.int16), np.zeros(2, np.int16)
    w, h, s = C.c_uint16(), C.c_uint16(), C.c_float()
    L.check(lib.fr_render_glyph_dims(L.ptr(b), units_per_em, font_size, L.ptr(mn), L.ptr(mx), C.byref(w), C.byref(h), C.byref(s)))
    return (int(mn[0]), int(mn[1])), (int(mx[0]), int(mx[1])), w.value, h.value, s.value


def renderGlyph(glyph: Glyph, font_info: FontInformation, font_size: int, *, ctx: Optional[Context] = None,
                mode: int = L.FR_GRAY_DEBUG) -> Gray:
    """render_glyph.zig:11 — `pub fn renderGlyph(glyph, font_info, font_size) !Image.Gray`."""
    ctx = ctx or default_context()
    _, _, w, h, _ = render_glyph_dims(glyph.box.as_array(), font_info.units_per_em, font_size)
    im = Gray.init(w, h)                                                     # :22
    pts, cstart, nc = _flat(glyph)
    box = glyph.box.as_array()
    L.check(ctx._lib.fr_render_glyph(ctx._h, L.ptr(pts), L.ptr(cstart), nc, L.ptr(box), font_info.units_per_em,
                                     font_size, mode, L.ptr(im.data)))
    return im


def renderGlyphWinding(glyph: Glyph, font_info: FontInformation, font_size: int, *, ctx: Optional[Context] = None,
                       scaler: int = 50, overflow_color: int = 150) -> Winding:
    """same grid as renderGlyph, raw i16 windings into an Image.Winding (Image.zig:85-130)"""
    ctx = ctx or default_context()
    _, _, w, h, _ = render_glyph_dims(glyph.box.as_array(), font_info.units_per_em, font_size)
    im = Winding.init(w, h, scaler, overflow_color)
    pts, cstart, nc = _flat(glyph)
    box = glyph.box.as_array()
    L.check(ctx._lib.fr_render_glyph(ctx._h, L.ptr(pts), L.ptr(cstart), nc, L.ptr(box), font_info.units_per_em,
                                     font_size, L.FR_WINDING_I16, L.ptr(im.data)))
    return im


class GlyphInfo:
    """render_glyph.zig:76-155.  contours[c] = list of (curve_type, include_p0)."""
    CURVE_TYPES = ("x_axis", "balance", "up_stright", "up_normal", "up_u", "up_inv_u",
                   "down_stright", "down_normal", "down_inv_u", "down_u")        # :84-95 enum order

    def __init__(self, curve_type: np.ndarray, include_p0: np.ndarray, curves_per_contour: Sequence[int]):
        self.curve_type, self.include_p0 = curve_type, include_p0
        self.contours, o = [], 0
        for n in curves_per_contour:
            self.contours.append(list(zip(curve_type[o:o + n].tolist(), include_p0[o:o + n].astype(bool).tolist())))
            o += n

    @staticmethod
    def init(glyph: Glyph, *, ctx: Optional[Context] = None) -> "GlyphInfo":    # :110
        ctx = ctx or default_context()
        pts, cstart, nc = _flat(glyph)
        n = glyph.curve_count
        ct, ip = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
        L.check(ctx._lib.fr_glyph_info_init(ctx._h, L.ptr(pts), L.ptr(cstart), nc, L.ptr(ct), L.ptr(ip)))
        return GlyphInfo(ct[:n], ip[:n], [c.curve_count for c in glyph.contours])

    def deinit(self) -> None:                                                    # :148
        self.contours = []


def windingInGlyph(glyph: Glyph, glyph_info: Optional[GlyphInfo], point, *, ctx: Optional[Context] = None):
    """render_glyph.zig:160.  `point` is one (x, y) or an (n, 2) int16 array; glyph_info is
    recomputed on the device (it is a pure function of the glyph) and only type-checked here."""
    ctx = ctx or default_context()
    q = np.ascontiguousarray(point, np.int16).reshape(-1, 2)
    pts, cstart, nc = _flat(glyph)
    out = np.zeros(len(q), np.int16)
    L.check(ctx._lib.fr_winding_in_glyph(ctx._h, L.ptr(pts), L.ptr(cstart), nc, L.ptr(q), len(q), L.ptr(out)))
    return int(out[0]) if np.ndim(point) == 1 else out


def winding_lattice(glyph: Glyph, *, ctx: Optional[Context] = None) -> np.ndarray:
    """the lattice of Image.GlyphDebug.render (Image.zig:220-240): (H, W) int16"""
    ctx = ctx or default_context()
    pts, cstart, nc = _flat(glyph)
    box = glyph.box.as_array()
    W, H = int(box[2]) - int(box[0]) + 3, int(box[3]) - int(box[1]) + 3
    out = np.zeros((H, W), np.int16)
    L.check(ctx._lib.fr_winding_lattice(ctx._h, L.ptr(pts), L.ptr(cstart), nc, L.ptr(box), L.ptr(out)))
    return out


def exact_lattice(glyph: Glyph, K: int, x0: int, y0: int, w: int, h: int, *, ctx: Optional[Context] = None) -> np.ndarray:
    """SURVEY §8 f-3 (build-defined): GlyphInfo.init + windingInGlyph (render_glyph.zig:110-300) on the
    glyph scaled by K, at the integer points (x0 + i, y0 - j) of the scaled glyph: (h, w) int16"""
    ctx = ctx or default_context()
    pts, cstart, nc = _flat(glyph)
    out = np.zeros((h, w), np.int16)
    L.check(ctx._lib.fr_exact_lattice(ctx._h, L.ptr(pts), L.ptr(cstart), nc, K, x0, y0, w, h, L.ptr(out)))
    return out


def exact_coverage(glyph: Glyph, K: int, x0: int, y0: int, w_px: int, h_px: int, n: int, *,
                   ctx: Optional[Context] = None) -> np.ndarray:
    """n x n lattice points per pixel of exact_lattice -> round_half_up(255 * inside / n^2): (h_px, w_px) u8"""
    ctx = ctx or default_context()
    pts, cstart, nc = _flat(glyph)
    out = np.zeros((h_px, w_px), np.uint8)
    L.check(ctx._lib.fr_exact_coverage(ctx._h, L.ptr(pts), L.ptr(cstart), nc, K, x0, y0, w_px, h_px, n, L.ptr(out)))
    return out


def glyph_debug_render(glyph: Glyph, winding_scale: int, *, ctx: Optional[Context] = None) -> RGB:
    """Image.GlyphDebug.render (Image.zig:220-240) through fr_glyph_debug_render: the exact-integer lattice
    coloured by setWindingLinear (:192-200) with the glyph's points marked (:202-218)"""
    ctx = ctx or default_context()
    pts, cs, nc = _flat(glyph)
    box = glyph.box.as_array()
    W, H = int(box[2]) - int(box[0]) + 3, int(box[3]) - int(box[1]) + 3
    out = np.zeros((H * W, 3), np.uint8)
    L.check(ctx._lib.fr_glyph_debug_render(ctx._h, L.ptr(pts), L.ptr(cs), nc, L.ptr(box), winding_scale, L.ptr(out)))
    return RGB(W, H, out)


def build_id() -> str:
    return L.load_library().fr_build_id().decode()
