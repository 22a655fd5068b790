"""Font — host-side mirror of /root/reference/src/font/Font.zig on the C ABI's contour producer
(fr_font_*: TrueType glyf/loca walk restated in C++, font-renderer_amd/csrc/fr_font.cpp).
Same surface as the reference: Font.initTTF(file) (:31), font.information (:25-29),
font.getGlyph(char) -> (Glyph, advance_width) (:161-169), plus glyph-index access and a
whole-font GlyphSet for batch rendering."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional

import numpy as np

from . import _lib as L
from .glyph import Box, Contour, FontInformation, Glyph, GlyphSet

FR_FONT_ALLOW_HINTED = 1


class Font:
    def __init__(self, data: bytes, allow_hinted: bool = False):
        self._lib = L.load_library()
        h = C.c_void_p()
        buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
        L.check(self._lib.fr_font_open(buf, len(data), FR_FONT_ALLOW_HINTED if allow_hinted else 0, C.byref(h)))
        self._h = h
        upm, ng, y0 = C.c_uint16(), C.c_uint16(), C.c_int()
        L.check(self._lib.fr_font_info(self._h, C.byref(upm), C.byref(ng), C.byref(y0)))
        self.information = FontInformation(upm.value, bool(y0.value))
        self.num_glyphs = ng.value

    @staticmethod
    def initTTF(path: str, allow_hinted: bool = False) -> "Font":       # Font.zig:31
        with open(path, "rb") as f:
            return Font(f.read(), allow_hinted)

    def glyph_index(self, char: int) -> int:                           # CharGlyphMapping.getGlyph
        gi = C.c_uint16()
        L.check(self._lib.fr_font_char_to_glyph(self._h, char, C.byref(gi)))
        return gi.value

    def glyph_by_index(self, gi: int) -> Glyph:                        # Font.zig:171 loadGlyph
        nc, npnt = C.c_uint32(), C.c_uint32()
        box = np.zeros(4, np.int16)
        L.check(self._lib.fr_font_glyph_measure(self._h, gi, C.byref(nc), C.byref(npnt), L.ptr(box)))
        pts = np.zeros((max(npnt.value, 1), 2), np.int16)
        cs = np.zeros(nc.value + 1, np.uint32)
        L.check(self._lib.fr_font_glyph_fill(self._h, gi, L.ptr(pts), L.ptr(cs)))
        contours = [Contour(pts[int(cs[c]):int(cs[c + 1])].copy()) for c in range(nc.value)]
        return Glyph(Box(int(box[0]), int(box[1]), int(box[2]), int(box[3])), contours)

    def advance_width(self, gi: int) -> int:                           # Font.advance_widths[gi] (Font.zig:123-139)
        a = C.c_int16()
        L.check(self._lib.fr_font_glyph_advance(self._h, gi, C.byref(a)))
        return a.value

    def getGlyph(self, char: int):                                     # Font.zig:161-169 -> struct {Glyph, i16}
        gi = self.glyph_index(char)
        return self.glyph_by_index(gi), self.advance_width(gi)

    def glyphset(self, glyph_indices: Optional[Iterable[int]] = None, skip_unsupported: bool = True):
        """-> (GlyphSet, kept glyph indices): the whole font (or the given indices) ready for
        fr_glyphset_create; glyphs the reference would panic on are skipped when asked."""
        idx = range(self.num_glyphs) if glyph_indices is None else glyph_indices
        glyphs: List[Glyph] = []
        kept: List[int] = []
        for gi in idx:
            try:
                glyphs.append(self.glyph_by_index(gi))
                kept.append(gi)
            except L.FrError:
                if not skip_unsupported:
                    raise
        return GlyphSet(glyphs), kept

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.fr_font_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
