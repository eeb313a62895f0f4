"""Import shim: the package directory is named ``mega-nerf-viewer_amd`` (with hyphens, after the
reference repository), which Python cannot import by name.  This module exposes it as
``mega_nerf_viewer_amd`` by pointing ``__path__`` at that directory and executing its
``__init__.py`` in this namespace."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mega-nerf-viewer_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__, "r", encoding="utf-8") as _f:
    exec(compile(_f.read(), __file__, "exec"))
