"""gams_amd -- MI355X (gfx950) engine behind the `gams` wave / sw / locate / anno hot path.

The product is libgams_gpu.so (hand-written HIP, C ABI in include/gams_gpu.h) plus the C++ host
layer in gams_amd/host; this package only binds them for tests and bench.py.
"""
import os as _os

# kernel arguments in device memory (see gams_gpu_create): must be in the environment before the HIP
# runtime initialises, which may happen before the library's own setenv when torch is imported first
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import _lib  # noqa: F401,E402

__all__ = ["_lib", "engine"]
