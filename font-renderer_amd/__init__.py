"""font-renderer_amd — MI355X-native glyph rasterizer (host-side mirror).

Python mirror of the reference's host interface for ONE path
(/root/reference/src/tools/render_glyph.zig + the Glyph/Image types either side of
it), sitting on the C ABI of include/fr_raster.h (libfr_raster.so, hand-written HIP
for gfx950).  There is no CPU path in this package: every compute call goes through
the shared library and raises if it (or a GPU) is missing.
"""
from .glyph import Box, Contour, FontInformation, Glyph, GlyphSet  # noqa: F401
from .image import RGB, Gray, GlyphDebug, Winding  # noqa: F401
from ._lib import (  # noqa: F401
    FR_COVERAGE_U8, FR_SDF_U8, FR_GRAY_DEBUG, FR_MASK_NONZERO, FR_SAMPLE_CENTER, FR_SAMPLE_CORNER,
    FR_WINDING_I16, FrError, Job, lib_path, load_library,
)
from .render_glyph import (  # noqa: F401
    Context, GlyphInfo, Plan, DeviceGlyphSet, renderGlyph, render_glyph_dims, windingInGlyph,
    winding_lattice, exact_lattice, exact_coverage, glyph_debug_render, build_id,
)
from .font import Font  # noqa: F401
