"""Import shim: the package directory is `font-renderer_amd/` (not a valid Python
identifier), so `import font_renderer_amd` resolves here and uses that directory
as its package path."""
import os as _os

__package__ = "font_renderer_amd"
__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "font-renderer_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f, _os
