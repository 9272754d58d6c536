"""geodiffuser_amd — MI355X-native (gfx950) implementation of GeoDiffuser's geometry-guided attention-sharing hot path.

Python host code mirrors the reference's operator interface for this path (same names, argument meaning and error
behaviour); all arithmetic of the path runs in hand-written HIP kernels behind the C ABI of include/geodiff_hip.h.
"""
from ._lib import GeodiffError, load as load_library  # noqa: F401

__all__ = ["GeodiffError", "load_library"]
