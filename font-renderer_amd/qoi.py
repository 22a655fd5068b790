"""qoi.saveRGB — mirror of /root/reference/src/tools/qoi.zig:25 on the C ABI (fr_qoi_*)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def saveRGB(image) -> bytes:
    """image: (h, w, 3) u8 RGB array, or a 2-D u8 gray array / Image.Gray (encoded as {v,v,v})"""
    lib = L.load_library()
    if hasattr(image, "as_2d"):
        image = image.as_2d()
    a = np.ascontiguousarray(image, np.uint8)
    h, w = a.shape[:2]
    out = np.zeros(lib.fr_qoi_bound(w, h), np.uint8)
    n = C.c_size_t()
    if a.ndim == 3:
        L.check(lib.fr_qoi_encode_rgb(L.ptr(a), w, h, L.ptr(out), out.size, C.byref(n)))
    else:
        L.check(lib.fr_qoi_encode_gray(L.ptr(a), w, h, w, L.ptr(out), out.size, C.byref(n)))
    return out[:n.value].tobytes()
