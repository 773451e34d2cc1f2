"""ctypes binding of libgams_gpu.so (the C ABI in include/gams_gpu.h).

There is no CPU fallback: if the shared library is missing or a call fails, an
exception is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libgams_gpu.so")

OK, EINVAL, ENODEV, ENOMEM, EHIP, ESHORT, EUNSUPPORTED, ESTATE = range(8)
WAVE_PEAKS, WAVE_DENSE = 1, 2


class GamsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gams_gpu error {code}: {msg}")
        self.code = code


class WaveParams(C.Structure):
    _fields_ = [("size", C.c_int32), ("step", C.c_int32), ("lag", C.c_uint32),
                ("threshold", C.c_float), ("influence", C.c_float)]


PEAK_DTYPE = np.dtype([("ctg", np.uint32), ("window", np.uint32), ("gc_count", np.uint32),
                       ("signal", np.int32)])
SW_ROW_DTYPE = np.dtype([("feature", np.uint32), ("type", np.int32), ("distance", np.int32),
                         ("start", np.int32), ("end", np.int32), ("gc_content", np.float32),
                         ("gc_mean", np.float32), ("gc_stddev", np.float32), ("gc_cv", np.float32)])

# name -> (restype, argtypes); exactly the entry points include/gams_gpu.h and include/gams_gpu_diag.h declare
_VP = C.c_void_p
_PP = C.POINTER(C.c_void_p)
PROTOTYPES = {
    "gams_gpu_create": (C.c_int, [C.c_int, _PP]),
    "gams_gpu_destroy": (None, [_VP]),
    "gams_gpu_last_error": (C.c_char_p, [_VP]),
    "gams_gpu_device_info": (C.c_int, [_VP, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "gams_gpu_release_cached": (C.c_int, [_VP, _VP]),
    "gams_gpu_sync": (C.c_int, [_VP]),
    "gams_gpu_timer_start": (C.c_int, [_VP]),
    "gams_gpu_timer_stop": (C.c_int, [_VP, C.POINTER(C.c_float)]),
    "gams_gpu_last_kernel_ms": (C.c_int, [_VP, C.POINTER(C.c_float)]),
    "gams_gpu_host_alloc": (C.c_int, [_VP, C.c_uint64, _PP]),
    "gams_gpu_host_free": (None, [_VP, _VP]),
    "gams_window_count": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "gams_seqset_create": (C.c_int, [_VP, C.c_uint32, _VP, _PP]),
    "gams_seqset_upload": (C.c_int, [_VP, _VP, C.c_uint32, _VP]),
    "gams_seqset_upload_all": (C.c_int, [_VP, _VP, _VP]),
    "gams_seqset_layout": (C.c_int, [_VP, _VP, _VP, C.POINTER(C.c_uint64)]),
    "gams_seqset_upload_image": (C.c_int, [_VP, _VP, _VP, C.c_uint64, C.c_uint64]),
    "gams_seqset_destroy": (None, [_VP, _VP]),
    "gams_wave_plan_create": (C.c_int, [_VP, _VP, C.POINTER(WaveParams), C.c_uint32, _PP]),
    "gams_wave_plan_destroy": (None, [_VP, _VP]),
    "gams_wave_total_windows": (C.c_uint64, [_VP]),
    "gams_wave_ctg_windows": (C.c_uint32, [_VP, C.c_uint32]),
    "gams_wave_run_n": (C.c_int, [_VP, _VP, C.c_uint32]),
    "gams_wave_plan_set_depth": (C.c_int, [_VP, _VP, C.c_uint32]),
    "gams_wave_plan_select": (C.c_int, [_VP, _VP, C.c_uint32]),
    "gams_wave_plan_set_lane": (C.c_int, [_VP, _VP, C.c_uint32]),
    "gams_wave_plan_set_taper": (C.c_int, [_VP, _VP, C.c_int]),
    "gams_wave_plan_set_taper_shape": (C.c_int, [_VP, _VP, C.c_int, C.c_int]),
    "gams_wave_plan_set_queue_threads": (C.c_int, [_VP, _VP, C.c_uint32]),
    "gams_wave_plan_kernel_name": (C.c_int, [_VP, _VP, C.c_char_p, C.c_size_t]),
    "gams_wave_plan_set_pipelined": (C.c_int, [_VP, _VP, C.c_int]),
    "gams_wave_run": (C.c_int, [_VP, _VP]),
    "gams_wave_peaks": (C.c_int, [_VP, _VP, _PP, C.POINTER(C.c_uint64)]),
    "gams_wave_rows_setup": (C.c_int, [_VP, _VP, C.POINTER(C.c_char_p), _VP, C.c_float]),
    "gams_wave_rows_begin": (C.c_int, [_VP, _VP]),
    "gams_wave_rows_end": (C.c_int, [_VP, _VP, _PP, C.POINTER(C.c_uint64), _PP]),
    "gams_wave_signal_text": (C.c_int, [_VP, _VP, C.POINTER(C.c_char_p), _VP, _PP, C.POINTER(C.c_uint64), _PP]),
    "gams_wave_dense": (C.c_int, [_VP, _VP, C.c_uint32, _VP, _VP]),
    "gams_wave_plan_set_tile": (C.c_int, [_VP, _VP, C.c_uint32]),
    "gams_wave_plan_set_threads": (C.c_int, [_VP, _VP, C.c_uint32]),
    "gams_wave_plan_settled": (C.c_int, [_VP, _VP, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "gams_wave_exact_count": (C.c_int, [_VP, _VP, C.POINTER(C.c_uint64)]),
    "gams_wave_plan_set_guard": (C.c_int, [_VP, _VP, C.c_float, C.c_int]),
    "gams_wave_plan_set_stamps": (C.c_int, [_VP, _VP, C.c_int]),
    "gams_wave_stamps": (C.c_int, [_VP, _VP, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "gams_wave_stamps_raw": (C.c_int, [_VP, _VP, _VP, C.c_uint64]),
    "gams_gpu_wave": (C.c_int, [_VP, _VP, C.c_uint32, C.POINTER(WaveParams), _VP, _VP, C.POINTER(C.c_uint32)]),
    "gams_gpu_sw": (C.c_int, [_VP, _VP, C.c_uint32, C.c_int32, _VP, _VP, C.c_uint32, C.c_int32, C.c_int32,
                              C.c_int32, _VP, C.c_uint64, C.POINTER(C.c_uint64)]),
    "gams_gpu_sw_batch": (C.c_int, [_VP, _VP, C.c_uint32, _VP, _VP, _VP, _VP, _VP, C.c_int32, C.c_int32, C.c_int32,
                                    _VP, C.c_uint64, _VP, _VP]),
    "gams_gpu_sw_text": (C.c_int, [_VP, _VP, C.c_uint32, _VP, C.POINTER(C.c_char_p), _VP, _VP, _VP, _VP, C.POINTER(C.c_char_p),
                                   C.c_int32, C.c_int32, C.c_int32, _PP, C.POINTER(C.c_uint64), _PP, C.POINTER(C.c_uint64)]),
    "gams_gpu_range_gc": (C.c_int, [_VP, _VP, C.c_uint32, C.c_int32, _VP, _VP, C.c_uint32, _VP]),
    "gams_gpu_range_gc_batch": (C.c_int, [_VP, _VP, C.c_uint32, _VP, _VP, _VP, _VP, _VP, _VP]),
    "gams_index_create": (C.c_int, [_VP, C.c_uint32, _VP, _VP, _VP, _PP]),
    "gams_index_destroy": (None, [_VP, _VP]),
    "gams_gpu_count": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_uint64, _VP]),
    "gams_gpu_locate": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_uint64, _VP]),
    "gams_spans_create": (C.c_int, [_VP, C.c_uint32, _VP, _VP, _VP, _PP]),
    "gams_spans_destroy": (None, [_VP, _VP]),
    "gams_gpu_cover": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_uint64, _VP]),
    "gams_gpu_valid_spans": (C.c_int, [_VP, _VP, C.c_uint64, C.c_int32, C.c_int32, _VP, _VP, C.c_uint64,
                                       C.POINTER(C.c_uint64)]),
}

_lib = None


def bind(path, strict=True):
    """dlopen one build of the library and bind every prototype of include/gams_gpu.h + gams_gpu_diag.h.
    strict=False (tools/ab.py comparing against an older build) skips entry points it lacks."""
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        if not strict and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """dlopen libgams_gpu.so and bind every prototype; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _lib = bind(SO_PATH)
    return _lib
