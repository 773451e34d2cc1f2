#!/usr/bin/env python3
"""Device-time rates of the secondary kernels (sw, locate --count, locate, anno) on synthetic
inputs shaped like BASELINE configs[2] / configs[4], per GPU share.  Kernel time only
(gams_gpu_last_kernel_ms: HIP events around the kernel); host<->device copies excluded."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

# optional argument: another build of libgams_gpu.so to measure (A/B of index layouts)
eng = engine.Engine(0, lib=_lib.bind(os.path.abspath(sys.argv[1]), strict=False)) if len(sys.argv) > 1 else engine.Engine(0)
lib = eng.lib


def kernel_ms():
    ms = C.c_float()
    eng.check(lib.gams_gpu_last_kernel_ms(eng.h, C.byref(ms)))
    return ms.value


rng = np.random.default_rng(5)

# ---- sw: 1e5 point features on a 30 Mb chromosome cut in 1-Mb ctgs (config 2 shape) ----
chrom = synth.chromosome(30_000_000, 9)
ctgs = synth.gen_ctgs("9", chrom, piece=1000000)
ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
nf = 100000 // len(ctgs)
feats = [np.sort(rng.integers(c["chr_start"], c["chr_end"] + 1, nf)).astype(np.int32) for c in ctgs]
for rep in range(2):                                   # one launch per ctg (gams_gpu_sw)
    rows_total, ms_total = 0, 0.0
    for i, c in enumerate(ctgs):
        n = C.c_uint64()
        rows = np.zeros(nf * 41, _lib.SW_ROW_DTYPE)
        eng.check(lib.gams_gpu_sw(eng.h, ss.p, i, c["chr_start"], feats[i].ctypes.data, feats[i].ctypes.data, nf, 100, 20,
                                  500, rows.ctypes.data, rows.size, C.byref(n)))
        rows_total += n.value
        ms_total += kernel_ms()
print(f"sw, one call per ctg ({len(ctgs)} launches): {rows_total} rows in {ms_total:.3f} ms kernel time -> "
      f"{rows_total / ms_total / 1e6:.2f} G rows/s, {rows_total * 84 / ms_total / 1e6:.1f} GB/s at 84 B/row")
sel = np.arange(len(ctgs), dtype=np.uint32)            # every ctg in one launch (gams_gpu_sw_batch)
cst = np.array([c["chr_start"] for c in ctgs], np.int32)
foff = (np.arange(len(ctgs) + 1) * nf).astype(np.uint64)
fall = np.ascontiguousarray(np.concatenate(feats), np.int32)
n = C.c_uint64()
rows = np.zeros(nf * 41 * len(ctgs), _lib.SW_ROW_DTYPE)
for rep in range(3):
    t0 = time.time()
    eng.check(lib.gams_gpu_sw_batch(eng.h, ss.p, sel.size, sel.ctypes.data, cst.ctypes.data, foff.ctypes.data,
                                    fall.ctypes.data, fall.ctypes.data, 100, 20, 500, rows.ctypes.data, rows.size, None,
                                    C.byref(n)))
    call_s = time.time() - t0
ms = kernel_ms()
print(f"sw, one batched call: {n.value} rows in {ms:.3f} ms kernel time -> {n.value / ms / 1e6:.2f} G rows/s, "
      f"{n.value * 84 / ms / 1e6:.1f} GB/s at 84 B/row; whole call (rows to pageable host memory) {call_s * 1e3:.1f} ms")

# ---- locate --count: 1.25e7 stored points + 1.25e7 queries over 4000 ctgs (config 4 per-GPU share) ----
n_ctg, per = 4000, 1_000_000
m = 12_500_000
g_of = np.sort(rng.integers(0, n_ctg, m)).astype(np.uint32)
off = np.searchsorted(g_of, np.arange(n_ctg + 1)).astype(np.uint64)
starts = (g_of.astype(np.uint64) % 32 * 0 + rng.integers(1, per, m)).astype(np.uint32)
stops = starts + 1
ix = C.c_void_p()
t0 = time.time()
eng.check(lib.gams_index_create(eng.h, n_ctg, off.ctypes.data, starts.ctypes.data, stops.ctypes.data, C.byref(ix)))
print(f"index_create ({m} intervals, {n_ctg} groups): {time.time() - t0:.2f} s upload + device sort + directories")
nq = 12_500_000
qg = rng.integers(0, n_ctg, nq).astype(np.uint32)
qs = rng.integers(1, per, nq).astype(np.uint32)
qe = qs + rng.integers(1, 2000, nq).astype(np.uint32)
out = np.zeros(nq, np.int32)
for name, order in (("unsorted queries", np.arange(nq)), ("queries sorted by (ctg, start)", np.lexsort((qs, qg)))):
    a, b, c = qg[order].copy(), qs[order].copy(), qe[order].copy()
    for _ in range(2):
        eng.check(lib.gams_gpu_count(eng.h, ix, a.ctypes.data, b.ctypes.data, c.ctypes.data, nq, out.ctypes.data))
    ms = kernel_ms()
    print(f"count, {name}: {nq / ms / 1e6:.2f} G queries/s, {nq * 16 / ms / 1e6:.1f} GB/s at 16 B/query ({ms:.3f} ms)")
hit = np.zeros(nq, np.int64)
for _ in range(2):
    eng.check(lib.gams_gpu_locate(eng.h, ix, qg.ctypes.data, qs.ctypes.data, qe.ctypes.data, nq, hit.ctypes.data))
ms = kernel_ms()
print(f"locate (first overlap), unsorted: {nq / ms / 1e6:.2f} G queries/s ({ms:.3f} ms)")
lib.gams_index_destroy(eng.h, ix)

# ---- anno: 1e6 spans per chr x 4 chr, 1.25e7 lines ----
n_chr, n_sp = 4, 1_000_000
lo_all, hi_all, soff = [], [], [0]
for _ in range(n_chr):
    cuts = np.sort(rng.choice(np.arange(1, 2_000_000_000, 97), 2 * n_sp, replace=False))
    lo_all.append(cuts[0::2].astype(np.int32))
    hi_all.append((cuts[1::2] - 1).astype(np.int32))
    soff.append(soff[-1] + n_sp)
lo = np.concatenate(lo_all)
hi = np.concatenate(hi_all)
soff = np.array(soff, np.uint64)
sp = C.c_void_p()
for rep in range(2):
    if sp:
        lib.gams_spans_destroy(eng.h, sp)
    t0 = time.time()
    eng.check(lib.gams_spans_create(eng.h, n_chr, soff.ctypes.data, lo.ctypes.data, hi.ctypes.data, C.byref(sp)))
    t_sp = time.time() - t0
print(f"spans_create ({lo.size} spans, {n_chr} groups): {t_sp * 1e3:.1f} ms")
g = rng.integers(0, n_chr, nq).astype(np.uint32)
s = rng.integers(1, 1_999_000_000, nq).astype(np.int32)
e = (s + rng.integers(0, 2000, nq)).astype(np.int32)
cl = (s - s % 1_000_000 + 1).astype(np.int32)
ch = (cl + 999_999).astype(np.int32)
prop = np.zeros(nq, np.float32)
for _ in range(2):
    eng.check(lib.gams_gpu_cover(eng.h, sp, g.ctypes.data, cl.ctypes.data, ch.ctypes.data, s.ctypes.data,
                                 e.ctypes.data, nq, prop.ctypes.data))
ms = kernel_ms()
print(f"anno cover: {nq / ms / 1e6:.2f} G lines/s, {nq * 16 / ms / 1e6:.1f} GB/s at 16 B/line ({ms:.3f} ms)")
lib.gams_spans_destroy(eng.h, sp)

# ---- call-level (host arrays in, host array out) rate of the query kernels ----
ix = C.c_void_p()
eng.check(lib.gams_index_create(eng.h, n_ctg, off.ctypes.data, starts.ctypes.data, stops.ctypes.data, C.byref(ix)))
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    eng.check(lib.gams_gpu_count(eng.h, ix, qg.ctypes.data, qs.ctypes.data, qe.ctypes.data, nq, out.ctypes.data))
    best = min(best, time.perf_counter() - t0)
print(f"count, whole call with pageable host arrays ({nq} queries): {best * 1e3:.2f} ms -> {nq / best / 1e9:.2f} G queries/s, "
      f"{nq * 16 / best / 1e9:.1f} GB/s over PCIe")


def pinned(n, dtype):
    p = C.c_void_p()
    eng.check(lib.gams_gpu_host_alloc(eng.h, n * np.dtype(dtype).itemsize, C.byref(p)))
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(p.value)
    return np.frombuffer(buf, dtype=dtype), p


pg, hg = pinned(nq, np.uint32)
ps, hs = pinned(nq, np.uint32)
pe, he = pinned(nq, np.uint32)
po, ho = pinned(nq, np.int32)
pg[:], ps[:], pe[:] = qg, qs, qe
best = 1e9
for _ in range(4):
    t0 = time.perf_counter()
    eng.check(lib.gams_gpu_count(eng.h, ix, pg.ctypes.data, ps.ctypes.data, pe.ctypes.data, nq, po.ctypes.data))
    best = min(best, time.perf_counter() - t0)
assert np.array_equal(po, out)
print(f"count, whole call with page-locked host arrays (gams_gpu_host_alloc): {best * 1e3:.2f} ms -> "
      f"{nq / best / 1e9:.2f} G queries/s, {nq * 16 / best / 1e9:.1f} GB/s over PCIe (12 B in + 4 B out per query)")
del pg, ps, pe, po
for hp in (hg, hs, he, ho):
    lib.gams_gpu_host_free(eng.h, hp)
lib.gams_index_destroy(eng.h, ix)
