"""CPU suite: the C-ABI library loads, exports every symbol include/fr_raster.h declares,
fails loudly without a GPU (no CPU fallback), and its host-only helper matches the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import font_renderer_amd as fr
from font_renderer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fr_raster.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    lib = fr.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in fr_raster.h but not exported"
    assert sorted(s[0] for s in _lib.SYMBOLS) == declared      # the ctypes table binds all of them
    assert lib.fr_abi_version() == 1


def test_job_struct_layout():
    assert C.sizeof(_lib.Job) == 32 and C.sizeof(_lib.RasterParams) == 16
    from font_renderer_amd.render_glyph import make_jobs
    j = make_jobs([(3, -5, 7, 16, 32, 48, 64, 0.5)])
    assert j.dtype.itemsize == 32
    raw = _lib.Job.from_buffer_copy(j.tobytes())
    assert (raw.glyph, raw.min_x, raw.max_y, raw.w, raw.h, raw.out_x, raw.out_y, raw.scale) == (3, -5, 7, 16, 32, 48, 64, 0.5)


def test_render_glyph_dims_matches_oracle(oracle, ascii_set):
    """fr_render_glyph_dims is host arithmetic (render_glyph.zig:13-19): no GPU needed"""
    for i in range(0, len(ascii_set), 7):
        for size in (9, 64, 200):
            box, upm = ascii_set.gs.boxes[i], int(ascii_set.g_upm[i])
            assert fr.render_glyph_dims(box, upm, size) == oracle.render_glyph_dims(box, upm, size)
    with pytest.raises(fr.FrError):
        fr.render_glyph_dims([0, 0, 30000, 30000], 16, 65535)      # leaves i16: the reference would trap


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(fr.FrError) as e:
        fr.Context(0)
    assert e.value.code == -2 and "no CPU path" in str(e.value)


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "font-renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in txt and "fr_oracle" not in txt and "libfr_oracle" not in txt, f
