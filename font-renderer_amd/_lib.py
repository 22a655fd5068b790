"""ctypes binding of include/fr_raster.h (libfr_raster.so).

Loading fails loudly: there is no Python / CPU implementation to fall back to."""
from __future__ import annotations

import ctypes as C
import os

FR_WINDING_I16, FR_GRAY_DEBUG, FR_MASK_NONZERO, FR_COVERAGE_U8, FR_SDF_U8 = 0, 1, 2, 3, 4
FR_SAMPLE_CORNER, FR_SAMPLE_CENTER = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path() -> str:
    # FR_RASTER_LIB: diagnostic builds only (stamps / ablation); the product is libfr_raster.so
    return os.environ.get("FR_RASTER_LIB") or os.path.join(_HERE, "libfr_raster.so")


class FrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"fr_raster error {code}: {msg}")
        self.code = code


class Job(C.Structure):          # == fr_job
    _fields_ = [("glyph", C.c_uint32), ("min_x", C.c_int32), ("max_y", C.c_int32),
                ("w", C.c_uint32), ("h", C.c_uint32), ("out_x", C.c_uint32), ("out_y", C.c_uint32),
                ("scale", C.c_float)]


class RasterParams(C.Structure):  # == fr_raster_params
    _fields_ = [("mode", C.c_int32), ("samples_per_axis", C.c_int32), ("sample_phase", C.c_int32),
                ("reserved", C.c_int32)]


# every symbol include/fr_raster.h declares: (name, restype, argtypes)
_P = C.c_void_p
_I16P, _U32P, _U8P = C.POINTER(C.c_int16), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
SYMBOLS = [
    ("fr_abi_version", C.c_int, []),
    ("fr_last_error", C.c_char_p, []),
    ("fr_build_id", C.c_char_p, []),
    ("fr_ctx_create", C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    ("fr_ctx_destroy", None, [_P]),
    ("fr_ctx_sync", C.c_int, [_P]),
    ("fr_ctx_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("fr_glyphset_create", C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_uint32, C.POINTER(_P)]),
    ("fr_glyphset_destroy", None, [_P]),
    ("fr_glyphset_prepare", C.c_int, [_P]),
    ("fr_glyphset_stats", C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("fr_plan_create", C.c_int, [_P, _P, _P, C.c_uint32, C.POINTER(RasterParams), C.POINTER(_P)]),
    ("fr_plan_destroy", None, [_P]),
    ("fr_plan_render", C.c_int, [_P, _P, C.c_size_t, C.c_size_t]),
    ("fr_plan_render_timed", C.c_int, [_P, _P, C.c_size_t, C.c_size_t, C.POINTER(C.c_float)]),
    ("fr_plan_pixels", C.c_uint64, [_P]),
    ("fr_plan_stats", C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("fr_plan_describe", C.c_int, [_P, C.c_char_p, C.c_size_t]),
    ("fr_allgather_bands", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("fr_gather_bands", C.c_int, [_P, _P, _P, C.c_size_t, C.c_int]),
    ("fr_render_batch", C.c_int, [_P, _P, _P, C.c_uint32, C.POINTER(RasterParams), _P, C.c_size_t, C.c_size_t]),
    ("fr_render_glyph_dims", C.c_int, [_P, C.c_uint16, C.c_uint16, _P, _P, C.POINTER(C.c_uint16),
                                       C.POINTER(C.c_uint16), C.POINTER(C.c_float)]),
    ("fr_render_glyph", C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_uint16, C.c_uint16, C.c_int32, _P]),
    ("fr_glyph_info_init", C.c_int, [_P, _P, _P, C.c_uint32, _P, _P]),
    ("fr_winding_in_glyph", C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_uint32, _P]),
    ("fr_winding_lattice", C.c_int, [_P, _P, _P, C.c_uint32, _P, _P]),
    ("fr_glyph_debug_render", C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_uint8, _P]),
    ("fr_atlas_layout", C.c_int, [_P, C.c_uint32, C.c_uint32, _P, C.c_uint32, C.c_uint16, C.c_uint32, C.c_uint32, C.c_uint32,
                                  _P, _P, C.POINTER(C.c_uint32)]),
    ("fr_atlas_layout_glyph_dims", C.c_int, [_P, C.c_uint32, C.c_uint32, _P, C.c_uint32, C.c_uint16, C.c_uint32, C.c_uint32, _P,
                                             C.POINTER(C.c_uint32)]),
    ("fr_exact_lattice", C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, _P]),
    ("fr_exact_coverage", C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, _P]),
    ("fr_font_open", C.c_int, [_P, C.c_size_t, C.c_uint32, C.POINTER(_P)]),
    ("fr_font_close", None, [_P]),
    ("fr_font_info", C.c_int, [_P, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.POINTER(C.c_int)]),
    ("fr_font_char_to_glyph", C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint16)]),
    ("fr_font_glyph_advance", C.c_int, [_P, C.c_uint16, C.POINTER(C.c_int16)]),
    ("fr_font_glyph_measure", C.c_int, [_P, C.c_uint16, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _P]),
    ("fr_font_glyph_fill", C.c_int, [_P, C.c_uint16, _P, _P]),
    ("fr_qoi_bound", C.c_size_t, [C.c_uint32, C.c_uint32]),
    ("fr_qoi_encode_rgb", C.c_int, [_P, C.c_uint32, C.c_uint32, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("fr_qoi_encode_gray", C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("fr_selftest_sqrt", C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    ("fr_selftest_division", C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
]

_lib = None


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own
    libamdhip64.so.7; if this library pulled in /opt/rocm's copy first, a later
    `import torch` would bring a second runtime into the process (observed: "No HIP GPUs
    are available" + a crash at exit).  So when torch is installed, its runtime is loaded
    first and libfr_raster.so binds to it by SONAME.  FR_HIP_RUNTIME=system skips this
    (pure C / Zig hosts never see torch and use /opt/rocm's runtime via RUNPATH)."""
    if os.environ.get("FR_HIP_RUNTIME", "") == "system":
        return
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library() -> C.CDLL:
    """dlopen libfr_raster.so (built by __graft_entry__.build()); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback.")
    _preload_hip_runtime()
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise FrError(rc, load_library().fr_last_error().decode("utf-8", "replace"))


def ptr(a):
    """numpy array -> void* (array must stay alive across the call)"""
    return a.ctypes.data_as(C.c_void_p)
