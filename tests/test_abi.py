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


def test_atlas_layout_matches_the_reference_formula(ascii_set):
    """fr_atlas_layout (host side of the C ABI): scale, floor / ceil in binary32 as render_glyph.zig:13-17,
    cell grid, pages — against the same formula written with numpy float32"""
    from font_renderer_amd.atlas import cell_jobs
    gs = ascii_set.gs
    n, cell, size, cols = 95, 128, 100, 16
    jobs, pages, n_pages = cell_jobs(gs, cell, size, ascii_set.g_upm, cols, first_glyph=95, n_glyphs=n, page_rows=2, return_pages=True)
    upm = ascii_set.g_upm[95:95 + n].astype(np.float32)
    scale = np.float32(size) / upm
    box = gs.boxes[95:95 + n].astype(np.float32)
    idx = np.arange(n)
    assert np.array_equal(jobs["glyph"], idx + 95)
    assert np.array_equal(jobs["scale"], scale)
    assert np.array_equal(jobs["min_x"], np.floor(box[:, 0] * scale).astype(np.int32))
    assert np.array_equal(jobs["max_y"], np.ceil(box[:, 3] * scale).astype(np.int32))
    assert (jobs["w"] == cell).all() and (jobs["h"] == cell).all()
    assert np.array_equal(jobs["out_x"], (idx % cols) * cell)
    assert np.array_equal(jobs["out_y"], ((idx % 32) // cols) * cell)       # two cell rows per page
    assert np.array_equal(pages, idx // 32) and n_pages == 3
    flat = cell_jobs(gs, 64, 50, 1000, 8, n_glyphs=20)                       # one units_per_em, no pages
    assert np.array_equal(flat["out_y"], (np.arange(20) // 8) * 64) and flat["scale"][0] == np.float32(50) / np.float32(1000)
    with pytest.raises(fr.FrError):
        cell_jobs(gs, 0, 50, 1000, 8, n_glyphs=4)


def test_qoi_writer_decodes_with_an_independent_decoder(oracle):
    """fr_qoi_encode_rgb / _gray (qoi.zig:25-88) decoded by Pillow's QOI plugin — a decoder this build did not
    write — and byte-identical to the oracle's restatement"""
    import io
    from PIL import Image as PILImage
    from font_renderer_amd import qoi
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (37, 29, 3), dtype=np.uint8)
    img[5:15] = img[5, 0]                       # runs longer than 62
    img[20, :, :] = (np.arange(29) * 3)[:, None]
    img[21:25] = img[21:25] // 64 * 64          # few colours: index hits
    stream = qoi.saveRGB(img)
    assert stream == oracle.qoi_encode(img)
    dec = np.asarray(PILImage.open(io.BytesIO(stream)).convert("RGB"))
    assert np.array_equal(dec, img)
    gray = rng.integers(0, 256, (19, 33), dtype=np.uint8)
    gray[4:9] = 200
    dec = np.asarray(PILImage.open(io.BytesIO(qoi.saveRGB(gray))).convert("RGB"))
    assert np.array_equal(dec, np.repeat(gray[:, :, None], 3, 2))


def _c_class(t: str) -> str:
    t = t.strip()
    if "*" in t or "[" in t:
        return "ptr"
    t = re.sub(r"\bconst\b", "", t).split()
    base = " ".join(w for w in t[:-1]) if len(t) > 1 else t[0]      # drop the parameter name
    base = base.strip() or t[0]
    return {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "uint16_t": "u16", "int16_t": "i16", "uint8_t": "u8",
            "uint64_t": "u64", "int64_t": "i64", "size_t": "usize", "float": "f32", "void": "void"}[base]


def _zig_class(t: str) -> str:
    t = t.strip()
    if t.startswith(("*", "?*", "[*", "?[*")):
        return "ptr"
    return {"c_int": "i32", "i32": "i32", "u32": "u32", "u16": "u16", "i16": "i16", "u8": "u8", "u64": "u64", "i64": "i64",
            "usize": "usize", "f32": "f32", "void": "void"}[t]


def _split_params(p: str):
    out, depth, cur = [], 0, ""
    for ch in p:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out if x.strip() and x.strip() != "void"]


def test_zig_binding_matches_the_header():
    """bindings/fr_raster.zig cannot be compiled here (no zig toolchain), so it is pinned textually: every function
    of include/fr_raster.h is declared `extern "c"` with the same name, the same number of parameters and the same
    integer / float / pointer classes (and return class); enums and the two structs carry the header's values / fields."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fr_raster.h")).read(), flags=re.S)
    zig = open(os.path.join(ROOT, "bindings", "fr_raster.zig")).read()
    c_fns = {}
    for m in re.finditer(r"^\s*((?:const\s+)?[A-Za-z_0-9]+\s*\*?)\s*(fr_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.M | re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        c_fns[name] = ("ptr" if "*" in ret else _c_class(ret + " x"), [_c_class(q) for q in _split_params(params)])
    z_fns = {}
    for m in re.finditer(r'pub extern "c" fn (fr_[a-z0-9_]+)\((.*?)\) ([^;]+);', zig):
        name, params, ret = m.group(1), m.group(2), m.group(3)
        z_fns[name] = (_zig_class(ret), [_zig_class(q.split(":", 1)[1]) for q in _split_params(params)])
    assert sorted(c_fns) == _declared_symbols()
    assert sorted(z_fns) == sorted(c_fns), sorted(set(c_fns) ^ set(z_fns))
    for name in c_fns:
        assert z_fns[name] == c_fns[name], (name, z_fns[name], c_fns[name])
    # enums
    def c_enum(tag):
        body = re.search(r"typedef enum %s \{(.*?)\}" % tag, hdr, flags=re.S).group(1)
        return [int(v) for v in re.findall(r"=\s*(-?\d+)", body)]
    def z_enum(name):
        body = re.search(r"pub const %s = enum\([a-z_0-9]+\) \{(.*?)\};" % name, zig, flags=re.S).group(1)
        return [int(v) for v in re.findall(r"=\s*(-?\d+)", body)]
    assert z_enum("Mode") == c_enum("fr_mode") == [0, 1, 2, 3, 4]
    assert z_enum("SamplePhase") == c_enum("fr_sample_phase")
    assert z_enum("Status") == c_enum("fr_status")
    # structs: same field names in the same order
    def c_fields(tag):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (tag, tag), hdr, flags=re.S).group(1)
        return [n for line in body.split(";") for n in re.findall(r"([a-z_0-9]+)\s*(?:,|$)", line.strip().split(" ", 1)[-1]) if line.strip()]
    def z_fields(name):
        body = re.search(r"pub const %s = extern struct \{(.*?)\};" % name, zig, flags=re.S).group(1)
        return re.findall(r"([a-z_0-9]+):", body)
    assert z_fields("Job") == ["glyph", "min_x", "max_y", "w", "h", "out_x", "out_y", "scale"]
    assert set(c_fields("fr_job")) == set(z_fields("Job"))
    assert z_fields("RasterParams") == ["mode", "samples_per_axis", "sample_phase", "reserved"]
