"""versatilefilmgrain_amd -- MI355X (gfx950) film grain synthesis hardware layer.

The product is the C-ABI shared library ``libvfgs_hip.so`` (include/vfgs_hip.h).  This
package only holds its sources (csrc/), the in-tree build recipe (build.py) and a thin
ctypes mirror of the C interface (hw.py).  There is no CPU fallback: importing ``hw`` without
the built library, or calling it without a gfx950 device, fails loudly.
"""
from .build import LIB  # noqa: F401
from .build import build as build_library  # noqa: F401

__all__ = ["build_library", "LIB"]
