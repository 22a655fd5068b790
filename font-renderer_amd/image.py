"""Image sinks — mirror of /root/reference/src/tools/Image.zig:44-130 (Gray, Winding).
Row-major `data[y*width + x]`; getRGBLinear reproduces the reference's colour maps."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class Gray:                     # Image.zig:44-83
    width: int
    height: int
    data: np.ndarray            # (height*width,) u8

    @staticmethod
    def init(width: int, height: int) -> "Gray":            # Image.zig:58-61
        return Gray(width, height, np.zeros(width * height, np.uint8))

    def getWidth(self) -> int:
        return self.width

    def getHeight(self) -> int:
        return self.height

    def getRGBLinear(self, index: int):                     # Image.zig:78-82
        v = int(self.data[index])
        return (v, v, v)

    def as_2d(self) -> np.ndarray:
        return self.data.reshape(self.height, self.width)


@dataclass
class Winding:                  # Image.zig:85-130
    width: int
    height: int
    data: np.ndarray            # (height*width,) i16
    scaler: int = 50
    overflow_color: int = 150

    @staticmethod
    def init(width: int, height: int, scaler: int, overflow_color: int) -> "Winding":   # Image.zig:101-104
        return Winding(width, height, np.zeros(width * height, np.int16), scaler, overflow_color)

    def getWidth(self) -> int:
        return self.width

    def getHeight(self) -> int:
        return self.height

    def getRGBLinear(self, index: int):                     # Image.zig:121-129
        val = int(self.data[index])
        if val == 0:
            return (0, 0, 0)
        c = min(self.scaler * abs(val), 65535)              # u16 saturating multiply
        color = min(c, 255)
        sub = 0 if c == color else self.overflow_color
        return (sub, sub, color) if val > 0 else (color, sub, sub)

    def as_2d(self) -> np.ndarray:
        return self.data.reshape(self.height, self.width)


@dataclass
class RGB:                      # Image.zig:132-170
    width: int
    height: int
    data: np.ndarray            # (height*width, 3) u8

    def getWidth(self) -> int:
        return self.width

    def getHeight(self) -> int:
        return self.height

    def getRGBLinear(self, index: int):
        return tuple(int(v) for v in self.data[index])

    def as_3d(self) -> np.ndarray:
        return self.data.reshape(self.height, self.width, 3)


@dataclass
class GlyphDebug:               # Image.zig:173-241
    """Image.GlyphDebug: the exact-integer winding lattice (1 px per font unit, 1-unit border) coloured by
    setWindingLinear with the glyph's points marked; `render` runs on the device through fr_glyph_debug_render."""
    rgb: RGB
    winding_scale: int
    overflow_color: int = 150
    on_curve_color: tuple = (255, 255, 0)
    off_curve_color: tuple = (0, 255, 255)

    @staticmethod
    def render(glyph, winding_scale: int, *, ctx=None) -> "GlyphDebug":       # Image.zig:220
        from . import render_glyph as rg
        return GlyphDebug(rg.glyph_debug_render(glyph, winding_scale, ctx=ctx), winding_scale)
