"""Loads tests/golden/ascii_glyphs.npz (numpy.load, no pickle) into Glyph objects."""
import os

import numpy as np

from font_renderer_amd.glyph import Box, Contour, Glyph, GlyphSet

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class AsciiSet:
    def __init__(self, z):
        self.z = z
        self.font_names = [str(s) for s in z["font_names"]]
        self.g_font, self.g_char, self.g_upm = z["g_font"], z["g_char"], z["g_upm"]
        self.gs = GlyphSet.from_arrays(z["exp_pts"], z["exp_cstart"], z["g_cont_start"], z["g_box"])

    def __len__(self):
        return len(self.g_char)

    def glyph(self, i) -> Glyph:
        return self.gs.glyph(i)

    def find(self, font_substr: str, ch: str) -> int:
        fi = [k for k, n in enumerate(self.font_names) if font_substr in n][0]
        hit = np.nonzero((self.g_font == fi) & (self.g_char == ord(ch)))[0]
        return int(hit[0])

    def raw(self, i):
        z = self.z
        c0, c1 = int(z["g_coord_start"][i]), int(z["g_coord_start"][i + 1])
        e0, e1 = int(z["g_end_start"][i]), int(z["g_end_start"][i + 1])
        return z["raw_coords"][c0:c1], z["raw_on"][c0:c1], z["raw_end"][e0:e1]


def load_ascii() -> AsciiSet:
    return AsciiSet(np.load(os.path.join(GOLDEN, "ascii_glyphs.npz"), allow_pickle=False))
