"""gams_amd -- MI355X (gfx950) engine behind the `gams` wave / sw / locate / anno hot path.

The product is libgams_gpu.so (hand-written HIP, C ABI in include/gams_gpu.h) plus the C++ host
layer in gams_amd/host; this package only binds them for tests and bench.py.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "engine"]
