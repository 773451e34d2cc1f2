#!/usr/bin/env python3
"""locate --count kernel rate against the size of the index (does the working set fit the Infinity Cache?)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import engine  # noqa: E402

eng = engine.Engine(0)
lib = eng.lib
rng = np.random.default_rng(11)
per = 1_000_000
nq = 12_500_000
for n_ctg in (250, 500, 1000, 2000, 4000, 8000):
    m = n_ctg * 3125
    g_of = np.sort(rng.integers(0, n_ctg, m)).astype(np.uint32)
    off = np.searchsorted(g_of, np.arange(n_ctg + 1)).astype(np.uint64)
    starts = rng.integers(1, per, m).astype(np.uint32)
    stops = starts + 1
    ix = C.c_void_p()
    eng.check(lib.gams_index_create(eng.h, n_ctg, off.ctypes.data, starts.ctypes.data, stops.ctypes.data, C.byref(ix)))
    qg = rng.integers(0, n_ctg, nq).astype(np.uint32)
    qs = rng.integers(1, per, nq).astype(np.uint32)
    qe = qs + rng.integers(1, 2000, nq).astype(np.uint32)
    out = np.zeros(nq, np.int32)
    for _ in range(2):
        eng.check(lib.gams_gpu_count(eng.h, ix, qg.ctypes.data, qs.ctypes.data, qe.ctypes.data, nq, out.ctypes.data))
    ms = C.c_float()
    eng.check(lib.gams_gpu_last_kernel_ms(eng.h, C.byref(ms)))
    print(f"{m:9d} intervals in {n_ctg:5d} ctgs ({m * 32 / 1e6:6.0f} MB of bucket records): {nq / ms.value / 1e6:6.2f} G queries/s "
          f"({ms.value:.3f} ms)")
    lib.gams_index_destroy(eng.h, ix)
