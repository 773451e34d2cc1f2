#!/usr/bin/env python3
"""Large random run of the interval kernels against closed forms in numpy (Lapper::count, first overlap in
(start, stop) order, anno coverage) on groups of very different sizes and key distributions."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import engine  # noqa: E402

eng = engine.Engine(0)
lib = eng.lib
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nq_total = 0
for rd in range(rounds):
    rng = np.random.default_rng(9000 + rd)
    n_groups = int(rng.integers(1, 200))
    sizes = rng.choice([0, 1, 2, 5, 50, 1000, 20000], n_groups)
    starts, stops = [], []
    for n in sizes:
        kind = rng.integers(0, 4)
        if kind == 0:
            a = rng.integers(0, 1_000_000, n)
        elif kind == 1:
            a = rng.integers(0, 50, n) + 777_000                     # heavy duplicates
        elif kind == 2:
            a = (rng.integers(0, 3000, n) ** 2) % 4_000_000_000       # lumpy, wide range
        else:
            a = np.cumsum(rng.integers(0, 3, n))                     # long runs of equal / consecutive starts
        a = a.astype(np.uint32)
        starts.append(a)
        stops.append(a + rng.choice([1, 1, 2, 50, 5000], n).astype(np.uint32))
    off = np.cumsum([0] + [int(x) for x in sizes]).astype(np.uint64)
    S = np.concatenate(starts).astype(np.uint32) if off[-1] else np.zeros(0, np.uint32)
    T = np.concatenate(stops).astype(np.uint32) if off[-1] else np.zeros(0, np.uint32)
    ix = C.c_void_p()
    eng.check(lib.gams_index_create(eng.h, n_groups, off.ctypes.data, S.ctypes.data, T.ctypes.data, C.byref(ix)))
    nq = 200_000
    g = rng.integers(0, n_groups + 2, nq).astype(np.uint32)          # some unknown groups
    pick = np.concatenate([S, T, [0, 1, 5]]).astype(np.int64)
    qs = (rng.choice(pick, nq) + rng.integers(-3, 4, nq)).clip(0, 2**32 - 10)
    qe = (qs + rng.choice([0, 1, 2, 60, 6000, 10**7], nq)).clip(0, 2**32 - 10)
    qs, qe = qs.astype(np.uint32), qe.astype(np.uint32)
    cnt = np.zeros(nq, np.int32)
    hit = np.zeros(nq, np.int64)
    eng.check(lib.gams_gpu_count(eng.h, ix, g.ctypes.data, qs.ctypes.data, qe.ctypes.data, nq, cnt.ctypes.data))
    eng.check(lib.gams_gpu_locate(eng.h, ix, g.ctypes.data, qs.ctypes.data, qe.ctypes.data, nq, hit.ctypes.data))
    lib.gams_index_destroy(eng.h, ix)
    for k in range(n_groups):
        sel = np.flatnonzero(g == k)
        st, sp = starts[k], stops[k]
        ss, tt = np.sort(st), np.sort(sp)
        exp = np.searchsorted(ss, qe[sel], "left").astype(np.int64) - np.searchsorted(tt, qs[sel].astype(np.uint64) + 1, "left")
        assert np.array_equal(cnt[sel], exp), ("count", rd, k)
        # first overlap in (start, stop) order: the sorted pairs' first index with start < qe and stop > qs
        order = np.lexsort((sp, st))
        ls, lt = st[order], sp[order]
        h = hit[sel]
        none = h < 0
        j = (h[~none] - int(off[k])).astype(np.int64)
        got_s, got_t = st[j], sp[j]
        assert np.all((got_s < qe[sel][~none]) & (got_t > qs[sel][~none])), ("locate overlap", rd, k)
        # nothing earlier in (start, stop) order overlaps: check on a sample (exact, O(n) each)
        for q in np.flatnonzero(~none)[:40]:
            ov = np.flatnonzero((ls < qe[sel][q]) & (lt > qs[sel][q]))
            assert (ls[ov[0]], lt[ov[0]]) == (st[h[q] - int(off[k])], sp[h[q] - int(off[k])]), ("locate first", rd, k)
        for q in np.flatnonzero(none)[:40]:
            assert not np.any((ls < qe[sel][q]) & (lt > qs[sel][q])), ("locate none", rd, k)
    assert np.all(cnt[g >= n_groups] == 0) and np.all(hit[g >= n_groups] == -1)
    # ---- spans / cover ----
    n_sets = int(rng.integers(1, 20))
    los, his = [], []
    for _ in range(n_sets):
        m = int(rng.choice([0, 1, 3, 500, 30000]))
        cuts = np.sort(rng.choice(np.arange(-2_000_000, 2_000_000, 3), 2 * m, replace=False)) if m else np.zeros(0, np.int64)
        los.append(cuts[0::2].astype(np.int32))
        his.append((cuts[1::2] - 1).astype(np.int32))
    soff = np.cumsum([0] + [x.size for x in los]).astype(np.uint64)
    LO = np.concatenate(los).astype(np.int32) if soff[-1] else np.zeros(0, np.int32)
    HI = np.concatenate(his).astype(np.int32) if soff[-1] else np.zeros(0, np.int32)
    sp_ = C.c_void_p()
    eng.check(lib.gams_spans_create(eng.h, n_sets, soff.ctypes.data, LO.ctypes.data, HI.ctypes.data, C.byref(sp_)))
    gg = rng.integers(0, n_sets + 1, nq).astype(np.uint32)
    s = rng.integers(-2_100_000, 2_100_000, nq).astype(np.int32)
    e = (s + rng.choice([0, 1, 99, 3000, 500000], nq)).astype(np.int32)
    cl = (s - rng.choice([0, 5, 10**5, 10**7], nq)).astype(np.int32)
    ch = (e + rng.choice([-2, 0, 7, 10**7], nq)).astype(np.int32)
    prop = np.zeros(nq, np.float32)
    eng.check(lib.gams_gpu_cover(eng.h, sp_, gg.ctypes.data, cl.ctypes.data, ch.ctypes.data, s.ctypes.data, e.ctypes.data,
                                 nq, prop.ctypes.data))
    lib.gams_spans_destroy(eng.h, sp_)
    for k in range(n_sets):
        sel = np.flatnonzero(gg == k)
        lo, hi = los[k].astype(np.int64), his[k].astype(np.int64)
        cum = np.concatenate([[0], np.cumsum(hi - lo + 1)])

        def upto(x):            # covered positions <= x
            i = np.searchsorted(lo, x, "right")
            out = cum[i].copy()
            has = i > 0
            over = np.zeros_like(x)
            over[has] = np.maximum(hi[i[has] - 1] - x[has], 0)
            return out - over

        L = np.maximum(s[sel], cl[sel]).astype(np.int64)
        H = np.minimum(e[sel], ch[sel]).astype(np.int64)
        card = np.where(H >= L, upto(H) - upto(L - 1), 0)
        exp = (card.astype(np.int32).astype(np.float32) / (e[sel].astype(np.int64) - s[sel] + 1).astype(np.float32))
        assert np.array_equal(prop[sel], exp.astype(np.float32)), ("cover", rd, k)
    assert np.all(prop[gg >= n_sets] == 0.0)
    nq_total += 3 * nq
    print(f"round {rd}: {n_groups} groups / {int(off[-1])} intervals, {n_sets} span sets / {int(soff[-1])} spans ok", flush=True)
print(f"interval fuzz: {rounds} rounds, {nq_total} queries (count, locate, cover), all equal to the closed forms")
