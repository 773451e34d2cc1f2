"""Interval kernels through the C ABI on key distributions that stress the bucket directories:
duplicates, one cluster plus an outlier, keys at 0 and 2^32-2, empty groups, negative span coordinates.
Expected values: the oracle (rust-lapper / anno semantics) on every query."""
import ctypes as C

import numpy as np
import pytest

from gams_amd import engine
from oracle import oracle as ora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = engine.Engine(0)
    yield e
    e.close()


def _groups(rng):
    """list of (starts, stops) per group, unsorted, stop > start"""
    out = []
    out.append((np.zeros(0, np.uint32), np.zeros(0, np.uint32)))                      # empty group
    out.append((np.array([7], np.uint32), np.array([8], np.uint32)))                  # single interval
    a = np.full(500, 1000, np.uint32)                                                 # all starts equal
    out.append((a, a + rng.integers(1, 50, 500).astype(np.uint32)))
    a = np.concatenate([rng.integers(10, 20, 300), [4_000_000_000]]).astype(np.uint32)   # cluster + far outlier
    out.append((a, a + 1))
    a = np.concatenate([[0, 0, 1], rng.integers(0, 2**32 - 3000, 2000), [2**32 - 2]]).astype(np.uint32)
    out.append((a, (a.astype(np.uint64) + rng.integers(1, 2000, a.size)).clip(max=2**32 - 1).astype(np.uint32)))
    a = rng.integers(1, 1_000_000, 5000).astype(np.uint32)                            # the uniform case
    out.append((a, a + rng.choice([1, 1, 1, 6, 300], 5000).astype(np.uint32)))
    a = (np.arange(3000, dtype=np.uint32) * 2) ** 2 % 100_000 + 5                       # lumpy
    out.append((a, a + 3))
    out.append((np.zeros(0, np.uint32), np.zeros(0, np.uint32)))                      # empty group at the end
    return out


def _queries(rng, groups, n_each):
    g_all, s_all, e_all = [], [], []
    for g, (st, sp) in enumerate(groups):
        edges = np.concatenate([st, sp, st - (st > 0), sp - 1, [0, 1, 2**32 - 2]]).astype(np.int64)
        s = np.concatenate([rng.choice(edges, n_each), rng.integers(0, 2**32 - 1, n_each // 4)])
        e = s + rng.choice([0, 1, 2, 50, 5000, 10**9], s.size)
        s = s.clip(0, 2**32 - 2)
        e = e.clip(0, 2**32 - 2)
        g_all.append(np.full(s.size, g))
        s_all.append(s)
        e_all.append(e)
    g_all.append(np.full(8, len(groups) + 3))                                         # unknown group
    s_all.append(np.arange(8))
    e_all.append(np.arange(8) + 5)
    order = rng.permutation(sum(x.size for x in g_all))
    return (np.concatenate(g_all)[order].astype(np.uint32), np.concatenate(s_all)[order].astype(np.uint32),
            np.concatenate(e_all)[order].astype(np.uint32))


def test_count_and_locate_on_skewed_keys(eng):
    rng = np.random.default_rng(77)
    groups = _groups(rng)
    off = np.cumsum([0] + [g[0].size for g in groups]).astype(np.uint64)
    starts = np.concatenate([g[0] for g in groups]).astype(np.uint32)
    stops = np.concatenate([g[1] for g in groups]).astype(np.uint32)
    ix = C.c_void_p()
    eng.check(eng.lib.gams_index_create(eng.h, len(groups), off.ctypes.data, starts.ctypes.data, stops.ctypes.data,
                                        C.byref(ix)))
    qg, qs, qe = _queries(rng, groups, 600)
    cnt = np.full(qg.size, -99, np.int32)
    hit = np.full(qg.size, -99, np.int64)
    eng.check(eng.lib.gams_gpu_count(eng.h, ix, qg.ctypes.data, qs.ctypes.data, qe.ctypes.data, qg.size,
                                     cnt.ctypes.data))
    eng.check(eng.lib.gams_gpu_locate(eng.h, ix, qg.ctypes.data, qs.ctypes.data, qe.ctypes.data, qg.size,
                                      hit.ctypes.data))
    eng.lib.gams_index_destroy(eng.h, ix)
    for g, (st, sp) in enumerate(groups):
        sel = np.flatnonzero(qg == g)
        ss, tt = np.sort(st), np.sort(sp)
        # Lapper::count in closed form, then the oracle itself on a sample
        exp = (np.searchsorted(ss, qe[sel], "left").astype(np.int64)
               - np.searchsorted(tt, qs[sel].astype(np.uint64) + 1, "left"))
        assert np.array_equal(cnt[sel], exp), g
        order = np.lexsort((sp, st))
        ls, lt = st[order], sp[order]
        for q in sel[:150]:
            assert cnt[q] == ora.lapper_count(ss, tt, int(qs[q]), int(qe[q]))
            k = ora.lapper_find_first(ls, lt, int(qs[q]), int(qe[q]))
            if k < 0:
                assert hit[q] == -1
            else:
                # equal (start, stop) pairs are interchangeable: compare the interval, not the index
                j = int(hit[q]) - int(off[g])
                assert (st[j], sp[j]) == (ls[k], lt[k]), (g, q)
    unknown = qg >= len(groups)
    assert np.all(cnt[unknown] == 0) and np.all(hit[unknown] == -1)


def test_cover_on_skewed_and_negative_spans(eng):
    rng = np.random.default_rng(78)
    sets = []
    sets.append((np.zeros(0, np.int32), np.zeros(0, np.int32)))
    sets.append((np.array([-2_000_000_000], np.int32), np.array([-1_999_999_000], np.int32)))
    cuts = np.sort(rng.choice(np.arange(-50_000, 50_000), 3000, replace=False))
    sets.append((cuts[0::2].astype(np.int32), (cuts[1::2] - 1).astype(np.int32)))
    lo = np.concatenate([np.arange(100, 400, 3), [2_000_000_000]]).astype(np.int32)      # cluster + outlier
    sets.append((lo, (lo + np.concatenate([np.ones(100, np.int32), [100]])).astype(np.int32)))
    keep = [s[1] >= s[0] for s in sets]
    sets = [(s[0][k], s[1][k]) for s, k in zip(sets, keep)]
    off = np.cumsum([0] + [s[0].size for s in sets]).astype(np.uint64)
    lo = np.concatenate([s[0] for s in sets]).astype(np.int32)
    hi = np.concatenate([s[1] for s in sets]).astype(np.int32)
    sp = C.c_void_p()
    eng.check(eng.lib.gams_spans_create(eng.h, len(sets), off.ctypes.data, lo.ctypes.data, hi.ctypes.data,
                                        C.byref(sp)))
    n = 6000
    g = rng.integers(0, len(sets) + 1, n).astype(np.uint32)                            # len(sets) = chr not in the set
    s = np.empty(n, np.int64)
    for i in range(n):
        pool = np.concatenate([sets[g[i]][0], sets[g[i]][1], [0]]) if g[i] < len(sets) else np.array([0])
        s[i] = int(rng.choice(pool)) + int(rng.integers(-3, 4))
    e = s + rng.choice([0, 1, 99, 3000, 10**6], n)
    s = s.clip(-2**31 + 1, 2**31 - 2).astype(np.int32)
    e = e.clip(s, 2**31 - 2).astype(np.int32)
    cl = (s.astype(np.int64) - rng.choice([0, 10, 10**5, 10**9], n)).clip(-2**31, 2**31 - 1).astype(np.int32)
    ch = (e.astype(np.int64) + rng.choice([-1, 0, 10, 10**9], n)).clip(-2**31, 2**31 - 1).astype(np.int32)
    prop = np.full(n, -1, np.float32)
    eng.check(eng.lib.gams_gpu_cover(eng.h, sp, g.ctypes.data, cl.ctypes.data, ch.ctypes.data, s.ctypes.data,
                                     e.ctypes.data, n, prop.ctypes.data))
    eng.lib.gams_spans_destroy(eng.h, sp)
    for i in range(n):
        if g[i] >= len(sets):
            assert prop[i] == 0.0
            continue
        slo, shi = sets[g[i]]
        exp = ora.anno_prop(slo, shi, int(cl[i]), int(ch[i]), int(s[i]), int(e[i]))
        assert prop[i] == np.float32(exp), (i, g[i], s[i], e[i], cl[i], ch[i])


@pytest.mark.parametrize("n_handles", [1, 2, 3])
def test_count_and_cover_sharded_over_handles_equal_single(eng, n_handles):
    """SURVEY 8e for the interval path: groups split over the handles (LPT), queries routed to the owner, results
    scattered back -- identical to one handle holding everything."""
    from gams_amd import host

    rng = np.random.default_rng(500 + n_handles)
    groups = _groups(rng)
    off = np.cumsum([0] + [g[0].size for g in groups]).astype(np.uint64)
    starts = np.concatenate([g[0] for g in groups]).astype(np.uint32)
    stops = np.concatenate([g[1] for g in groups]).astype(np.uint32)
    qg, qs, qe = _queries(rng, groups, 400)
    engs = [eng] + [engine.Engine(0) for _ in range(n_handles - 1)]
    got = host.count_multi(engs, off, starts, stops, qg, qs, qe)
    ix = C.c_void_p()
    eng.check(eng.lib.gams_index_create(eng.h, len(groups), off.ctypes.data, starts.ctypes.data, stops.ctypes.data,
                                        C.byref(ix)))
    exp = np.zeros(qg.size, np.int32)
    eng.check(eng.lib.gams_gpu_count(eng.h, ix, qg.ctypes.data, qs.ctypes.data, qe.ctypes.data, qg.size, exp.ctypes.data))
    eng.lib.gams_index_destroy(eng.h, ix)
    assert np.array_equal(got, exp)
    # cover
    sets = []
    for k in range(7):
        m = int(rng.choice([0, 1, 40, 3000]))
        cuts = np.sort(rng.choice(np.arange(-100000, 100000, 2), 2 * m, replace=False)) if m else np.zeros(0, np.int64)
        sets.append((cuts[0::2].astype(np.int32), (cuts[1::2] - 1).astype(np.int32)))
    soff = np.cumsum([0] + [s[0].size for s in sets]).astype(np.uint64)
    lo = np.concatenate([s[0] for s in sets]).astype(np.int32)
    hi = np.concatenate([s[1] for s in sets]).astype(np.int32)
    n = 5000
    g = rng.integers(0, len(sets) + 1, n).astype(np.uint32)
    s = rng.integers(-101000, 101000, n).astype(np.int32)
    e = (s + rng.choice([0, 9, 500, 40000], n)).astype(np.int32)
    cl = (s - rng.choice([0, 100, 10**6], n)).astype(np.int32)
    ch = (e + rng.choice([-1, 0, 10**6], n)).astype(np.int32)
    got = host.cover_multi(engs, soff, lo, hi, g, cl, ch, s, e)
    sp = C.c_void_p()
    eng.check(eng.lib.gams_spans_create(eng.h, len(sets), soff.ctypes.data, lo.ctypes.data, hi.ctypes.data, C.byref(sp)))
    exp = np.zeros(n, np.float32)
    eng.check(eng.lib.gams_gpu_cover(eng.h, sp, g.ctypes.data, cl.ctypes.data, ch.ctypes.data, s.ctypes.data,
                                     e.ctypes.data, n, exp.ctypes.data))
    eng.lib.gams_spans_destroy(eng.h, sp)
    assert np.array_equal(got, exp)
    for x in engs[1:]:
        x.close()


def test_index_from_idx_blobs(eng):
    """idx:rg:{ctg} blobs (bincode of rust-lapper's Lapper, redis.rs:276-303) -> device index -> Lapper::count:
    the path a host takes that keeps the reference's Redis namespace."""
    from gams_amd import host

    rng = np.random.default_rng(91)
    groups = [(np.sort(rng.integers(1, 500000, n)).astype(np.uint32)) for n in (0, 1, 700, 5000)]
    blobs = [host.bincode_lapper(g, g + 1) for g in groups]
    ix = host.index_from_lappers(eng, blobs)
    qg = rng.integers(0, len(groups), 4000).astype(np.uint32)
    qs = rng.integers(1, 500000, 4000).astype(np.uint32)
    qe = qs + rng.integers(0, 3000, 4000).astype(np.uint32)
    cnt = np.full(4000, -1, np.int32)
    eng.check(eng.lib.gams_gpu_count(eng.h, ix, qg.ctypes.data, qs.ctypes.data, qe.ctypes.data, 4000, cnt.ctypes.data))
    eng.lib.gams_index_destroy(eng.h, ix)
    for g, st in enumerate(groups):
        sel = np.flatnonzero(qg == g)
        exp = np.searchsorted(st, qe[sel], "left") - np.searchsorted(st + 1, qs[sel].astype(np.int64) + 1, "left")
        assert np.array_equal(cnt[sel], exp), g


def test_host_alloc_blocks_are_pooled(eng):
    """gams_gpu_host_alloc / gams_gpu_host_free: a freed page-locked block is handed out again for the next request
    of that size (no second pinning), blocks can be freed in any order and are writable / readable."""
    import ctypes as C
    lib = eng.lib
    a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
    eng.check(lib.gams_gpu_host_alloc(eng.h, 3 << 20, C.byref(a)))
    eng.check(lib.gams_gpu_host_alloc(eng.h, 3 << 20, C.byref(b)))
    assert a.value and b.value and a.value != b.value
    C.memset(a, 0x5A, 3 << 20)
    C.memset(b, 0xA5, 3 << 20)
    assert C.string_at(a.value + (3 << 20) - 4, 4) == b"\x5a" * 4 and C.string_at(b, 4) == b"\xa5" * 4
    first = a.value
    lib.gams_gpu_host_free(eng.h, a)
    eng.check(lib.gams_gpu_host_alloc(eng.h, 3 << 20, C.byref(c)))
    assert c.value == first                                  # the block of before, not a new pinning
    lib.gams_gpu_host_free(eng.h, b)
    lib.gams_gpu_host_free(eng.h, c)
    lib.gams_gpu_host_free(eng.h, None)


def test_one_interval_group_wider_than_2_31(eng):
    """A group holding ONE interval whose stop - start has bit 31 set (idx:ctg of a one-ctg chromosome with
    absurd coordinates; the C ABI takes any u32): the count path's one-cell grid needs a shift of 32, which
    the directory builder has to reach in 64-bit arithmetic (a 32-bit `>> 32` never ends the loop
    on this hardware).  Expected values: Lapper semantics in closed form."""
    cases = [(0, 0x80000001), (5, 0xFFFFFFFE), (0x7FFFFFFF, 0xFFFFFFFF), (0, 0x7FFFFFFF)]
    starts = np.array([c[0] for c in cases], np.uint32)
    stops = np.array([c[1] for c in cases], np.uint32)
    off = np.arange(len(cases) + 1, dtype=np.uint64)
    ix = C.c_void_p()
    eng.check(eng.lib.gams_index_create(eng.h, len(cases), off.ctypes.data, starts.ctypes.data, stops.ctypes.data,
                                        C.byref(ix)))
    qs = np.array([0, 1, 0x7FFFFFFF, 0x80000000, 0x80000001, 0xFFFFFFF0, 4, 5], np.uint32)
    qe = (qs.astype(np.uint64) + np.array([1, 10, 2, 1, 5, 8, 1, 1])).clip(max=2**32 - 1).astype(np.uint32)
    for g, (a, b) in enumerate(cases):
        qg = np.full(qs.size, g, np.uint32)
        cnt = np.full(qs.size, -9, np.int32)
        hit = np.full(qs.size, -9, np.int64)
        eng.check(eng.lib.gams_gpu_count(eng.h, ix, qg.ctypes.data, qs.ctypes.data, qe.ctypes.data, qs.size,
                                         cnt.ctypes.data))
        eng.check(eng.lib.gams_gpu_locate(eng.h, ix, qg.ctypes.data, qs.ctypes.data, qe.ctypes.data, qs.size,
                                          hit.ctypes.data))
        # Lapper: an interval [a, b) overlaps the query [s, e) iff a < e and b > s
        exp = ((a < qe.astype(np.int64)) & (b > qs.astype(np.int64))).astype(np.int32)
        assert np.array_equal(cnt, exp), (g, cnt, exp)
        assert np.array_equal(hit, np.where(exp == 1, g, -1)), (g, hit)
        for q in range(qs.size):
            assert cnt[q] == ora.lapper_count(np.array([a], np.uint32), np.array([b], np.uint32), int(qs[q]), int(qe[q]))
    eng.lib.gams_index_destroy(eng.h, ix)


@pytest.mark.parametrize("sizes", [(0, 1, 2, 3, 255, 256, 257), (1024, 1025, 7), (2048, 2049), (4096, 4097, 0, 5), (8192, 31),
                                   (8193, 100), (20000, 3000, 1)])
def test_index_build_at_every_workgroup_capacity(eng, sizes):
    """The index build sorts a whole group in one workgroup's LDS (bitonic network, ties broken by the caller's order:
    the stable intervals.sort() of redis.rs:253,299) up to 8,192 intervals per group and hands larger ones to the
    library's segmented radix sort: groups of every size around the kernel's capacities, with many equal starts and
    equal (start, stop) pairs.  count against the closed form on all queries, locate against the oracle's scan
    (first hit in (start, stop) order; equal pairs are interchangeable)."""
    rng = np.random.default_rng(sum(sizes))
    groups = []
    for n in sizes:
        st = rng.integers(0, max(4, n // 3 + 1), n).astype(np.uint32) * 7 + 10           # ~3 intervals per start value
        sp = st + rng.choice([1, 1, 2, 50, 400], n).astype(np.uint32)
        groups.append((st, sp))
    off = np.cumsum([0] + list(sizes)).astype(np.uint64)
    starts = np.concatenate([g[0] for g in groups]).astype(np.uint32)
    stops = np.concatenate([g[1] for g in groups]).astype(np.uint32)
    ix = C.c_void_p()
    eng.check(eng.lib.gams_index_create(eng.h, len(groups), off.ctypes.data, starts.ctypes.data, stops.ctypes.data,
                                        C.byref(ix)))
    for g, (st, sp) in enumerate(groups):
        hi = int(st.max()) + 500 if st.size else 100
        qs = rng.integers(0, hi, 3000).astype(np.uint32)
        qe = (qs + rng.choice([1, 2, 9, 300, 5000], qs.size)).astype(np.uint32)
        qg = np.full(qs.size, g, np.uint32)
        cnt = np.full(qs.size, -9, np.int32)
        hit = np.full(qs.size, -9, np.int64)
        eng.check(eng.lib.gams_gpu_count(eng.h, ix, qg.ctypes.data, qs.ctypes.data, qe.ctypes.data, qs.size, cnt.ctypes.data))
        eng.check(eng.lib.gams_gpu_locate(eng.h, ix, qg.ctypes.data, qs.ctypes.data, qe.ctypes.data, qs.size, hit.ctypes.data))
        ss, tt = np.sort(st), np.sort(sp)
        exp = (np.searchsorted(ss, qe, "left").astype(np.int64) - np.searchsorted(tt, qs.astype(np.uint64) + 1, "left"))
        assert np.array_equal(cnt, exp), (g, sizes)
        assert np.array_equal(hit >= 0, exp > 0), (g, sizes)
        order = np.lexsort((sp, st))
        ls, lt = st[order], sp[order]
        for q in range(0, qs.size, 40):
            k = ora.lapper_find_first(ls, lt, int(qs[q]), int(qe[q]))
            if k < 0:
                assert hit[q] == -1
            else:
                j = int(hit[q]) - int(off[g])
                assert 0 <= j < st.size and (st[j], sp[j]) == (ls[k], lt[k]), (g, q)
    eng.lib.gams_index_destroy(eng.h, ix)
