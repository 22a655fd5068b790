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
