"""GPU parity of the wave path (through the C ABI) against the CPU oracle and the
reference's golden peaks.  Bit-exact: gc counts, signals, compacted peaks."""
import numpy as np
import pytest

import helpers
from gams_amd import _lib, engine
from oracle import oracle as ora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = engine.Engine(0)
    yield e
    e.close()


def synth(n, seed, gc=0.4, lower=0.2, nrate=1e-4):
    rng = np.random.default_rng(seed)
    x = np.arange(n)
    p = gc + 0.06 * np.sin(2 * np.pi * x / 2300) + 0.04 * np.sin(2 * np.pi * x / 97000)
    is_gc = rng.random(n) < p
    pick = rng.random(n) < 0.5
    s = np.where(is_gc, np.where(pick, ord("G"), ord("C")), np.where(pick, ord("A"), ord("T"))).astype(np.uint8)
    s = np.where(rng.random(n) < lower, s | 0x20, s).astype(np.uint8)
    s[rng.random(n) < nrate] = ord("N")
    return s


def check_dense(eng, seq, size, step, lag, thr, infl=1.0):
    cnt, sig = eng.wave(seq, size, step, lag, thr, infl)
    ocnt, _, osig = ora.wave_windows(seq, size, step, lag, thr, infl)
    assert cnt.size == ocnt.size
    assert np.array_equal(cnt, ocnt), f"gc_count mismatch at {np.flatnonzero(cnt != ocnt)[:5]}"
    bad = np.flatnonzero(sig.astype(np.int32) != osig)
    assert bad.size == 0, f"signal mismatch at {bad[:5]}: gpu {sig[bad[:5]]} oracle {osig[bad[:5]]}"
    return cnt, sig


def test_device_is_gfx950(eng):
    arch, cus, hbm = eng.device_info()
    assert arch.startswith("gfx950"), arch
    assert cus >= 200 and hbm > 200e9


# BASELINE config 0/1: yeast chr I, size 100 step 10 lag 100 thr 3 infl 1
def test_chrI_dense_matches_oracle(eng, s288c):
    cnt, sig = check_dense(eng, s288c["I"], 100, 10, 100, 3.0)
    assert cnt.size == 23012
    assert int((sig == 1).sum()) == 104 and int((sig == -1).sum()) == 278


@pytest.mark.parametrize("size,step,lag,thr", [
    (100, 1, 100, 3.0),    # config 4 geometry (step 1)
    (100, 5, 200, 3.0),    # doc/benchmark/Atha.md:383
    (100, 20, 50, 3.0),    # doc/benchmark/Atha.md:71
    (50, 7, 33, 2.5),      # size not a multiple of step
    (13, 3, 10, 2.0),
    (255, 10, 100, 3.0),
    (256, 10, 100, 3.0),   # gc counts need 16 bits
    (1000, 50, 30, 2.0),
    (100, 10, 2, 3.0),
    (100, 10, 1, 3.0),     # lag 1: std is NaN, never signals
    (100, 10, 600, 3.0),
    (100, 10, 700, 3.0),   # lag*size > 65535: 64-bit variance path
    (300, 10, 250, 3.0),   # 16-bit counts + 64-bit variance path
    (1, 1, 20, 1.0),
])
def test_param_sweep_mito(eng, s288c, size, step, lag, thr):
    check_dense(eng, s288c["Mito"], size, step, lag, thr)


@pytest.mark.parametrize("thr", [0.0, -1.0, float("inf"), float("nan"), 0.5, 10.0])
def test_threshold_edge_values(eng, s288c, thr):
    check_dense(eng, s288c["Mito"][:30000], 100, 10, 100, thr)


def test_low_complexity_and_n_runs(eng):
    # constant stretches make std == 0 or tiny: every decision sits in the guard band
    s = synth(200000, 7)
    s[50000:60000] = ord("N")
    s[90000:100000] = ord("A")
    s[120000:130000] = ord("G")
    s[150000:150050] = ord("N")
    period = np.frombuffer(b"ACGTTGCAAC", np.uint8)
    s[160000:170000] = np.tile(period, 1000)
    check_dense(eng, s, 100, 10, 100, 3.0)
    check_dense(eng, s, 100, 1, 100, 3.0)


def test_non_acgt_bytes_count_in_denominator_only(eng):
    rng = np.random.default_rng(3)
    s = rng.integers(0, 256, 100000, dtype=np.uint8)   # arbitrary bytes, incl. 0x00 and 0xFF
    check_dense(eng, s, 100, 10, 100, 3.0)


def test_ragged_lengths(eng):
    base = synth(70000, 11)
    for n in (1090, 1091, 1099, 1100, 1101, 4095, 4096, 4097, 40959, 40960, 40961, 65521):
        check_dense(eng, base[:n], 100, 10, 100, 3.0)


def test_short_ctg_is_an_error_like_the_reference_panic(eng):
    with pytest.raises(_lib.GamsError) as ei:
        eng.wave(b"ACGT" * 100, 100, 10, 100, 3.0, 1.0)     # 31 windows < lag 100
    assert ei.value.code == _lib.ESHORT
    with pytest.raises(ValueError):
        ora.wave_windows(b"ACGT" * 100, 100, 10, 100, 3.0, 1.0)


def test_influence_below_one_serial_path(eng, s288c):
    seq = s288c["I"][:120000]
    for infl in (0.0, 0.5, 0.9):
        check_dense(eng, seq, 100, 10, 100, 3.0, infl)


def peaks_of(osig, ocnt):
    idx = np.flatnonzero(osig != 0)
    return idx, ocnt[idx], osig[idx]


def test_batch_peaks_ordered_and_exact(eng, s288c):
    # several ctgs of ragged length in one launch; peaks must come back in (ctg, window) order
    seqs = [s288c["I"][:100000], s288c["I"][100000:], s288c["Mito"], synth(333333, 5), synth(5000, 6)]
    ss = engine.SeqSet(eng, seqs)
    for tile in (0, 256, 1024, 3072, 5120, 8192):
        plan = engine.WavePlan(eng, ss, 100, 10, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE,
                               tile_windows=tile)
        plan.run()
        pk = plan.peaks()
        exp = []
        for c, s in enumerate(seqs):
            ocnt, _, osig = ora.wave_windows(s, 100, 10, 100, 3.0, 1.0)
            cnt, sig = plan.dense(c)
            assert np.array_equal(cnt, ocnt) and np.array_equal(sig.astype(np.int32), osig)
            idx, k, sg = peaks_of(osig, ocnt)
            exp += [(c, int(i), int(kk), int(g)) for i, kk, g in zip(idx, k, sg)]
        got = [(int(r["ctg"]), int(r["window"]), int(r["gc_count"]), int(r["signal"])) for r in pk]
        assert got == exp
        assert plan.total_windows == sum(plan.ctg_windows(c) for c in range(len(seqs)))
        plan.close()
    ss.close()


def test_golden_peaks_rows_from_gpu_peaks(eng, s288c):
    """I.peaks.tsv rebuilt from the GPU's compacted peaks with the reference's merge rule
    (coverage <= 1: same-sign windows whose spans intersect chain together)."""
    seq = s288c["I"]
    ss = engine.SeqSet(eng, [seq])
    plan = engine.WavePlan(eng, ss, 100, 10, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    plan.run()
    pk = plan.peaks()
    rows = []
    for sgn in (1, -1):
        w = pk["window"][pk["signal"] == sgn].astype(np.int64)
        k = pk["gc_count"][pk["signal"] == sgn]
        i = 0
        while i < len(w):
            j = i
            while j + 1 < len(w) and (w[j + 1] - w[j]) * 10 < 100:
                j += 1
            s, e = 1 + w[i] * 10, w[j] * 10 + 100
            gc = ora.fmt_f32(float(np.float32(k[i]) / np.float32(100)))
            name = f"I(+):{s}-{e}" if j > i else f"I:{s}-{e}"
            rows.append((w[i], f"{name}\t{gc}\t{sgn}"))
            i = j + 1
    rows.sort()
    out = ["#range\tgc_content\tsignal"] + [r[1] for r in rows]
    assert out == helpers.read_lines("I.peaks.tsv")
    plan.close()
    ss.close()


def test_guard_band_is_rarely_taken(eng):
    s = synth(4_000_000, 21)
    ss = engine.SeqSet(eng, [s])
    plan = engine.WavePlan(eng, ss, 100, 10, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    plan.run()
    pk = plan.peaks()
    n_exact = plan.exact_count()
    ocnt, _, osig = ora.wave_windows(s, 100, 10, 100, 3.0, 1.0)
    idx = np.flatnonzero(osig)
    assert np.array_equal(pk["window"], idx) and np.array_equal(pk["signal"], osig[idx])
    assert np.array_equal(pk["gc_count"], ocnt[idx])
    assert n_exact < plan.total_windows * 2e-3, n_exact
    plan.close()
    ss.close()


def test_peak_slots_regrow_when_every_window_signals(eng, s288c):
    # threshold -1: |x - avg| > -std is always true, so each tile holds tile_windows peaks,
    # far beyond the default slot of tile_windows / 8 records
    seq = s288c["Mito"]
    ss = engine.SeqSet(eng, [seq, seq[:30000]])
    for tile in (0, 256, 3072):
        plan = engine.WavePlan(eng, ss, 100, 10, 100, -1.0, 1.0, flags=_lib.WAVE_PEAKS, tile_windows=tile)
        plan.run()
        pk = plan.peaks()
        exp = []
        for c, s in enumerate([seq, seq[:30000]]):
            ocnt, _, osig = ora.wave_windows(s, 100, 10, 100, -1.0, 1.0)
            idx = np.flatnonzero(osig)
            exp += [(c, int(i), int(ocnt[i]), int(osig[i])) for i in idx]
        got = [(int(r["ctg"]), int(r["window"]), int(r["gc_count"]), int(r["signal"])) for r in pk]
        assert got == exp and len(got) > 8000
        plan.run()                         # a second run after the regrow keeps working
        assert len(plan.peaks()) == len(exp)
        plan.close()
    ss.close()


def test_upload_all_ragged_batch_matches_per_ctg_content(eng):
    """gams_seqset_upload_all: zero-length ctgs, ctgs crossing and ending on the 16-MiB staging
    windows, and a later single-ctg replacement; content checked through range GC (utils.rs:141-162)."""
    import ctypes as C

    rng = np.random.default_rng(31)
    lens = [0, 1000, (16 << 20) + 3, 0, 300, (16 << 20) - 1000 - 256 * 2 - 512, 5000, 1, 0]
    alphabet = np.frombuffer(b"ACGTacgtNn", np.uint8)
    seqs = [alphabet[rng.integers(0, alphabet.size, n)].tobytes() for n in lens]
    ss = engine.SeqSet(eng, seqs)

    def check(i, seq):
        n = len(seq)
        if n == 0:
            return
        rs = np.concatenate([[1, 1, n], rng.integers(1, n + 1, 200)]).astype(np.int32)
        re = np.minimum(rs + rng.choice([0, 9, 99, 4999], rs.size), n).astype(np.int32)
        re[1] = n
        gc = np.zeros(rs.size, np.float32)
        eng.check(eng.lib.gams_gpu_range_gc(eng.h, ss.p, i, 1, rs.ctypes.data, re.ctypes.data, rs.size,
                                            gc.ctypes.data))
        for a, b, g in zip(rs, re, gc):
            assert g == np.float32(ora.range_gc_content(seq, 1, int(a), int(b))), (i, a, b)

    for i, s in enumerate(seqs):
        check(i, s)
    # gams_gpu_range_gc_batch: the ranges of several ctgs (skipped, repeated, out of order) in one launch
    sel = [6, 1, 4, 1, 7]
    per, rs_all, re_all = [], [], []
    for i in sel:
        n = len(seqs[i])
        rs = rng.integers(1, n + 1, 50).astype(np.int32)
        re = np.minimum(rs + rng.choice([0, 9, 99], rs.size), n).astype(np.int32)
        gc = np.zeros(rs.size, np.float32)
        eng.check(eng.lib.gams_gpu_range_gc(eng.h, ss.p, i, 1, rs.ctypes.data, re.ctypes.data, rs.size, gc.ctypes.data))
        per.append(gc)
        rs_all.append(rs)
        re_all.append(re)
    sel_a, cst = np.array(sel, np.uint32), np.ones(len(sel), np.int32)
    roff = (np.arange(len(sel) + 1) * 50).astype(np.uint64)
    rs_c, re_c = np.ascontiguousarray(np.concatenate(rs_all)), np.ascontiguousarray(np.concatenate(re_all))
    gc = np.zeros(rs_c.size, np.float32)
    eng.check(eng.lib.gams_gpu_range_gc_batch(eng.h, ss.p, len(sel), sel_a.ctypes.data, cst.ctypes.data, roff.ctypes.data,
                                              rs_c.ctypes.data, re_c.ctypes.data, gc.ctypes.data))
    assert np.array_equal(gc, np.concatenate(per))
    re_c[60] = len(seqs[1]) + 5                                         # a range outside its ctg is reported
    assert eng.lib.gams_gpu_range_gc_batch(eng.h, ss.p, len(sel), sel_a.ctypes.data, cst.ctypes.data, roff.ctypes.data,
                                           rs_c.ctypes.data, re_c.ctypes.data, gc.ctypes.data) == _lib.EINVAL
    new = alphabet[rng.integers(0, 4, lens[2])].tobytes()
    ss.upload(2, new)
    check(2, new)
    check(1, seqs[1])
    check(4, seqs[4])
    ss.close()


@pytest.mark.parametrize("depth", [2, 3, 4])
def test_depth_ways_hold_independent_identical_results(eng, s288c, depth):
    """gams_wave_plan_set_depth: runs rotate over `depth` streams / output sets; every held run
    equals the one-at-a-time result, also after an upload that the auxiliary streams must wait for."""
    seqs = [bytes(s288c["I"]), bytes(s288c["Mito"])]
    ss = engine.SeqSet(eng, seqs)
    ref_plan = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE)
    ref_plan.run()
    ref = ref_plan.peaks()
    ref_dense = [ref_plan.dense(i) for i in range(2)]
    plan = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE)
    plan.set_depth(depth)
    for _ in range(2 * depth + 1):
        plan.run()
    for age in range(depth):
        plan.select(age)
        assert np.array_equal(plan.peaks(), ref), age
        for i in range(2):
            cnt, sig = plan.dense(i)
            assert np.array_equal(cnt, ref_dense[i][0]) and np.array_equal(sig, ref_dense[i][1])
    with pytest.raises(_lib.GamsError):
        plan.select(depth)
    # new bases for ctg 1: every way has to pick them up
    mito2 = seqs[1][::-1]
    ss.upload(1, mito2)
    exp_cnt, _, exp_sig = ora.wave_windows(mito2, 100, 10, 100, 3.0, 1.0)
    for _ in range(depth):
        plan.run()
    for age in range(depth):
        plan.select(age)
        cnt, sig = plan.dense(1)
        assert np.array_equal(cnt, exp_cnt) and np.array_equal(sig, exp_sig), age
        pk = plan.peaks()
        mine = pk[pk["ctg"] == 1]
        idx = np.flatnonzero(exp_sig)
        assert np.array_equal(mine["window"], idx) and np.array_equal(mine["signal"], exp_sig[idx])
    plan.set_depth(1)
    plan.run()
    assert np.array_equal(plan.dense(1)[0], exp_cnt)
    plan.close()
    ref_plan.close()
    ss.close()


@pytest.mark.parametrize("depth,n", [(1, 40), (2, 40), (3, 100), (4, 257)])
def test_run_n_equals_repeated_run(eng, s288c, depth, n):
    """gams_wave_run_n (two queueing host threads from depth 3 on) leaves the same held passes as n calls of
    gams_wave_run; counters of every pass are intact."""
    seqs = [bytes(s288c["I"]), synth(200000, 11).tobytes()]
    ss = engine.SeqSet(eng, seqs)
    plan = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS)
    plan.run()
    ref = plan.peaks()
    ref_exact = plan.exact_count()
    plan.set_depth(depth)
    for rep in range(3):                       # several batches: the launcher thread is reused
        plan.run_n(n)
        for age in range(depth):
            plan.select(age)
            assert np.array_equal(plan.peaks(), ref), (rep, age)
            assert plan.exact_count() == ref_exact
    plan.run()                                  # mixing single runs with batches keeps the rotation
    plan.run_n(5)
    assert np.array_equal(plan.peaks(), ref)
    plan.close()
    ss.close()


@pytest.mark.parametrize("prm", [(100, 10, 100, 3.0), (100, 1, 100, 3.0), (50, 7, 33, 2.0), (255, 8, 60, 1.0)])
@pytest.mark.parametrize("tile", [1024, 2048, 3072, 5120, 7168])
def test_every_fast_tile_size_matches_the_oracle(eng, s288c, prm, tile):
    """W = 4 / 8 / 12 / 20 / 28 windows per thread (tile 1024 / 2048 / 3072 / 5120 / 7168; W = 28 exists for
    the baked step-1 parameters only, elsewhere 7168 is a tile of the general kernel), baked and run-time parameters."""
    seqs = [bytes(s288c["I"]), synth(33333, 3).tobytes(), bytes(s288c["Mito"][:20000])]
    ss = engine.SeqSet(eng, seqs)
    try:
        plan = engine.WavePlan(eng, ss, *prm, 1.0, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE, tile_windows=tile)
    except _lib.GamsError as e:
        assert e.code == _lib.EUNSUPPORTED
        ss.close()
        return
    plan.run()
    pk = plan.peaks()
    for c, s in enumerate(seqs):
        ocnt, _, osig = ora.wave_windows(s, *prm, 1.0)
        cnt, sig = plan.dense(c)
        assert np.array_equal(cnt, ocnt) and np.array_equal(sig.astype(np.int32), osig), (prm, tile, c)
        mine = pk[pk["ctg"] == c]
        idx = np.flatnonzero(osig)
        assert np.array_equal(mine["window"], idx) and np.array_equal(mine["signal"], osig[idx])
    plan.close()
    ss.close()


@pytest.mark.parametrize("size,step,lag,thr,infl", [
    (1000, 500, 100, 3.0, 1.0),      # coarse scan: tile bytes beyond 64 KB
    (300, 150, 200, 2.0, 1.0),
    (100, 700, 50, 2.5, 1.0),        # step > size: gaps between windows
    (50, 3, 30000, 3.0, 1.0),        # prefix arrays beyond the LDS
    (1000, 500, 100, 2.0, 0.5),      # + the serial recurrence
    (65535, 300, 2, 1.0, 1.0),       # the largest window
])
def test_untiled_path_for_halos_beyond_a_tile(eng, size, step, lag, thr, infl):
    """(lag+1)*step + size + 256*step > 64 KB or LDS overflow: wave_direct_*_kernel; same answers."""
    need = size + (lag + 40) * step
    seqs = [synth(need + 7777, 21).tobytes(), synth(need + 123, 22, gc=0.55).tobytes()]
    ss = engine.SeqSet(eng, seqs)
    plan = engine.WavePlan(eng, ss, size, step, lag, thr, infl, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE)
    plan.run()
    pk = plan.peaks()
    for c, s in enumerate(seqs):
        ocnt, _, osig = ora.wave_windows(s, size, step, lag, thr, infl)
        cnt, sig = plan.dense(c)
        assert np.array_equal(cnt, ocnt) and np.array_equal(sig.astype(np.int32), osig), (size, step, lag, c)
        mine = pk[pk["ctg"] == c]
        idx = np.flatnonzero(osig)
        assert np.array_equal(mine["window"], idx) and np.array_equal(mine["signal"], osig[idx])
        assert np.array_equal(mine["gc_count"], ocnt[idx])
    # peaks-only plan (no DENSE requested) goes the same way
    p2 = engine.WavePlan(eng, ss, size, step, lag, thr, infl, flags=_lib.WAVE_PEAKS)
    p2.run()
    assert np.array_equal(p2.peaks(), pk)
    p2.close()
    plan.close()
    ss.close()


def test_block_pool_is_reused_and_can_be_released(eng, s288c):
    """freed HBM / pinned blocks stay on the handle for the next batch; gams_gpu_release_cached drops them"""
    import ctypes as C

    held = C.c_uint64()
    eng.check(eng.lib.gams_gpu_release_cached(eng.h, C.byref(held)))
    for _ in range(3):
        ss = engine.SeqSet(eng, [bytes(s288c["I"])])
        plan = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS)
        plan.run()
        n = plan.peaks().size
        plan.close()
        ss.close()
    eng.check(eng.lib.gams_gpu_release_cached(eng.h, C.byref(held)))
    assert n > 0 and held.value >= len(s288c["I"])          # at least the sequence block was cached
    eng.check(eng.lib.gams_gpu_release_cached(eng.h, C.byref(held)))
    assert held.value == 0
    check_dense(eng, s288c["Mito"][:30000], 100, 10, 100, 3.0)   # and the handle keeps working


def test_out_of_memory_is_reported_and_survivable(eng, s288c):
    """a seqset beyond the HBM: GAMS_ENOMEM, and the handle keeps working afterwards"""
    import ctypes as C

    lens = np.full(120, 4_000_000_000, np.uint32)        # 480 GB
    p = C.c_void_p()
    rc = eng.lib.gams_seqset_create(eng.h, lens.size, lens.ctypes.data, C.byref(p))
    assert rc == _lib.ENOMEM, rc
    assert b"hipMalloc" in eng.lib.gams_gpu_last_error(eng.h)
    check_dense(eng, s288c["Mito"][:30000], 100, 10, 100, 3.0)


def test_call_order_and_argument_errors(eng, s288c):
    """status codes instead of panics: results before a run, wrong outputs for the plan's flags, bad indices"""
    import ctypes as C

    ss = engine.SeqSet(eng, [bytes(s288c["Mito"][:30000])])
    plan = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS)
    pk, n = C.c_void_p(), C.c_uint64()
    assert eng.lib.gams_wave_peaks(eng.h, plan.p, C.byref(pk), C.byref(n)) == _lib.ESTATE       # nothing ran yet
    assert eng.lib.gams_wave_plan_select(eng.h, plan.p, 0) == _lib.ESTATE
    plan.run()
    cnt = np.zeros(4000, np.uint32)
    assert eng.lib.gams_wave_dense(eng.h, plan.p, 0, cnt.ctypes.data, None) == _lib.ESTATE      # no DENSE output
    assert eng.lib.gams_wave_plan_set_depth(eng.h, plan.p, 0) == _lib.EINVAL
    assert eng.lib.gams_wave_plan_set_depth(eng.h, plan.p, 5) == _lib.EINVAL
    assert eng.lib.gams_wave_plan_select(eng.h, plan.p, 1) == _lib.ESTATE                       # depth 1 holds one run
    assert eng.lib.gams_seqset_upload(eng.h, ss.p, 3, cnt.ctypes.data) == _lib.EINVAL           # ctg index
    prm = _lib.WaveParams(0, 10, 100, 3.0, 1.0) if hasattr(_lib, "WaveParams") else None
    if prm is not None:
        bad = C.c_void_p()
        assert eng.lib.gams_wave_plan_create(eng.h, ss.p, C.byref(prm), _lib.WAVE_PEAKS, C.byref(bad)) == _lib.EINVAL
    assert plan.peaks().size > 0                                                               # still fine
    plan.close()
    ss.close()


def test_set_tile_that_leaves_the_tiled_kernels(eng):
    """a PEAKS-only plan whose requested tile pushes it to the untiled kernels (prefix arrays
    beyond the 160-KB LDS: size 100, step 1, lag 7000 is tiled at the default tile, untiled at 8192) must
    get the dense rows those kernels write.  Compared with the oracle either way."""
    seq = synth(60000, 21).tobytes()
    ss = engine.SeqSet(eng, [seq])
    plan = engine.WavePlan(eng, ss, 100, 1, 7000, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    ocnt, _, osig = ora.wave_windows(seq, 100, 1, 7000, 3.0, 1.0)
    idx = np.flatnonzero(osig)
    for tile in (0, 8192, 4096, 8192):
        if tile:
            eng.check(eng.lib.gams_wave_plan_set_tile(eng.h, plan.p, tile))
        plan.run()
        pk = plan.peaks()
        assert np.array_equal(pk["window"], idx) and np.array_equal(pk["signal"], osig[idx]), tile
        assert np.array_equal(pk["gc_count"], ocnt[idx]), tile
    plan.close()
    ss.close()


def test_plans_on_lanes_overlap_and_keep_their_results(eng, s288c):
    """gams_wave_plan_set_lane: three plans over three different batches on three streams of the handle
    (bench.py's rotation); every plan's peaks equal those of the same plan run alone on lane 0."""
    seqs = [[bytes(s288c["I"])], [synth(300000, 31).tobytes(), synth(70000, 32).tobytes()],
            [bytes(s288c["Mito"]), synth(150000, 33).tobytes()]]
    sets = [engine.SeqSet(eng, s) for s in seqs]
    plans = [engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS) for ss in sets]
    ref = []
    for p in plans:
        p.run()
        ref.append(p.peaks())
    for j, p in enumerate(plans):
        p.set_lane(j + 1)
    for rep in range(50):
        for p in plans:
            p.run()
    for j, p in enumerate(plans):
        assert np.array_equal(p.peaks(), ref[j]), j
    with pytest.raises(_lib.GamsError):
        plans[0].set_lane(4)
    plans[1].set_lane(0)
    plans[1].set_depth(3)
    plans[1].run_n(30)
    assert np.array_equal(plans[1].peaks(), ref[1])
    for p in plans:
        p.close()
    for ss in sets:
        ss.close()


def test_guard_band_margin_on_small_inputs(eng, s288c):
    """gams_wave_plan_set_guard: safety 1.0 (the derived bound with no factor on top) and all_exact (every
    window in the reference's f32 order) give the oracle's signals, like the default 1.5."""
    seqs = [bytes(s288c["I"]), bytes(s288c["Mito"]), synth(120000, 41).tobytes()]
    ss = engine.SeqSet(eng, seqs)
    plan = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE)
    total = plan.total_windows
    exact = {}
    for name, safety, all_exact in (("default", 1.5, False), ("safety1", 1.0, False), ("exact", 1.5, True),
                                    ("wide", 50.0, False)):
        plan.set_guard(safety, all_exact)
        plan.run()
        for c, s in enumerate(seqs):
            ocnt, _, osig = ora.wave_windows(s, 100, 10, 100, 3.0, 1.0)
            cnt, sig = plan.dense(c)
            assert np.array_equal(cnt, ocnt) and np.array_equal(sig.astype(np.int32), osig), (name, c)
        exact[name] = plan.exact_count()
    assert exact["safety1"] <= exact["default"] <= exact["wide"] <= exact["exact"]
    assert exact["default"] < 1e-3 * total
    # every window that can signal (i >= lag, per ctg) went through the exact path, bar those the
    # constant-count table settles (all lag counts and the window's own equal)
    assert 0.98 * (total - 100 * len(seqs)) <= exact["exact"] <= total - 100 * len(seqs)
    with pytest.raises(_lib.GamsError):
        plan.set_guard(0.5, False)
    plan.close()
    ss.close()


def test_reupload_is_ordered_behind_queued_passes(eng, s288c):
    """an upload into a live seqset queues behind the passes already queued on every stream of
    the handle, so a held pass keeps the results of the bytes it was queued on: run (old), upload, run (new);
    select(1) = old content, select(0) = new content."""
    a = synth(400000, 51).tobytes()
    b = synth(400000, 52).tobytes()
    ss = engine.SeqSet(eng, [a, bytes(s288c["Mito"])])
    plan = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS)
    plan.set_depth(2)
    exp = {}
    for name, seq in (("a", a), ("b", b)):
        _, _, sig = ora.wave_windows(seq, 100, 10, 100, 3.0, 1.0)
        exp[name] = np.flatnonzero(sig)
    for rep in range(5):
        ss.upload(0, a)
        plan.run_n(7)          # old content, several passes deep on both streams
        ss.upload(0, b)        # must not overtake them
        plan.run()
        plan.select(1)
        pk = plan.peaks()
        assert np.array_equal(pk[pk["ctg"] == 0]["window"], exp["a"]), rep
        plan.select(0)
        pk = plan.peaks()
        assert np.array_equal(pk[pk["ctg"] == 0]["window"], exp["b"]), rep
    plan.close()
    ss.close()


def test_overflowing_tile_regrows_to_the_fullest_tile(eng, s288c):
    """threshold -1 signals every window from `lag` on: the first peaks() regrows the slots (to the fullest
    tile, not beyond), the held history survives, later runs do not regrow again."""
    seq = bytes(s288c["Mito"])
    ss = engine.SeqSet(eng, [seq, synth(50000, 61).tobytes()])
    plan = engine.WavePlan(eng, ss, 100, 10, 100, -1.0, 1.0, flags=_lib.WAVE_PEAKS)
    plan.set_depth(2)
    plan.run_n(5)
    n = [plan.ctg_windows(0), plan.ctg_windows(1)]
    for age in (0, 1):
        plan.select(age)
        pk = plan.peaks()
        assert pk.size == sum(x - 100 for x in n), age
        assert np.array_equal(pk[pk["ctg"] == 0]["window"], np.arange(100, n[0]))
    plan.run_n(3)
    assert plan.peaks().size == sum(x - 100 for x in n)
    plan.close()
    ss.close()


@pytest.mark.parametrize("lag,infl,thr", [(100, 0.5, 3.0), (64, 0.0, 2.0), (65, 0.25, 2.5), (7, 0.9, 1.5), (200, 1.5, 3.0),
                                          (128, -0.5, 2.0)])
def test_influence_recurrence_one_wave_per_ctg(eng, s288c, lag, infl, thr):
    """influence != 1 (stat.rs:42): the filtered[] recurrence (round 2: one wavefront per ctg with an LDS ring; round 3:
    speculate-and-repair, one lane per zone); several ctgs of different lengths in one batch, lags around the 64-lane
    chunk size, influences outside [0, 1]."""
    seqs = [bytes(s288c["Mito"][:40000]), synth(25000, 71).tobytes(), synth(lag * 10 + 99, 72).tobytes(),
            bytes(s288c["I"][:60000])]
    ss = engine.SeqSet(eng, seqs)
    plan = engine.WavePlan(eng, ss, 100, 10, lag, thr, infl, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE)
    plan.run()
    pk = plan.peaks()
    for c, s in enumerate(seqs):
        ocnt, _, osig = ora.wave_windows(s, 100, 10, lag, thr, infl)
        cnt, sig = plan.dense(c)
        assert np.array_equal(cnt, ocnt)
        bad = np.flatnonzero(sig.astype(np.int32) != osig)
        assert bad.size == 0, (c, bad[:5])
        idx = np.flatnonzero(osig)
        assert np.array_equal(pk[pk["ctg"] == c]["window"], idx)
    plan.close()
    ss.close()


def test_plan_kernel_name_follows_the_plan(eng):
    """gams_wave_plan_kernel_name spells the instantiation the plan launches (the row rocprofv3 prints)."""
    seq = synth(300_000, 77)
    ss = engine.SeqSet(eng, [seq])
    names = {}
    for prm in [(100, 10, 100, 1.0), (100, 1, 100, 1.0), (50, 7, 33, 1.0), (100, 10, 100, 0.5), (100, 1000, 100, 1.0)]:
        plan = engine.WavePlan(eng, ss, prm[0], prm[1], prm[2], 3.0, prm[3], flags=_lib.WAVE_PEAKS)
        names[prm] = plan.kernel_name()
        if prm == (100, 10, 100, 1.0):
            plan.set_tile(3072)
            names["headline, 3072"] = plan.kernel_name()
        if prm == (100, 1, 100, 1.0):
            for tw in (5120, 7168):
                plan.set_tile(tw)
                names[f"step 1, {tw}"] = plan.kernel_name()
            plan.set_threads(256)
            names["step 1, 7168, four waves"] = plan.kernel_name()
        plan.close()
    ss.close()
    # a 300-kb ctg is a handful of tiles: the smallest tile, which is baked for the headline parameters only
    assert names[(100, 10, 100, 1.0)] == "wave_fast_kernel<4, 100, 10, 100, false, 256>"
    assert names["headline, 3072"] == "wave_fast_kernel<12, 100, 10, 100, false, 256>"
    assert names[(100, 1, 100, 1.0)] == "wave_fast_kernel<4, 0, 0, 0, false, 256>"
    assert names["step 1, 5120"] == "wave_fast_kernel<20, 100, 1, 100, false, 256>"
    assert names["step 1, 7168"] == "wave_fast_kernel<28, 100, 1, 100, false, 64>"       # W = 28: a tile per wave
    assert names["step 1, 7168, four waves"] == "wave_fast_kernel<28, 100, 1, 100, false, 256>"
    assert names[(50, 7, 33, 1.0)] == "wave_fast_kernel<4, 0, 0, 0, false, 256>"
    assert names[(100, 10, 100, 0.5)] == "jac_eval_kernel"
    assert names[(100, 1000, 100, 1.0)].startswith("wave_direct_count_kernel")


@pytest.mark.parametrize("step", [1, 5, 10, 20])
def test_size_and_step_baked_lag_as_argument(eng, s288c, step):
    """size 100 with step 1 / 5 / 10 / 20 and ANY lag keeps the baked kernel (lag is an argument there; the
    reference's own benchmark runs 100 / 5 / 200 and 100 / 20 / 50, doc/benchmark/Atha.md:55,276-280): counts,
    signals and peaks against the oracle for lags around every boundary of the tile layout, at every tile size
    that has such a kernel, on ragged ctgs."""
    pool = [bytes(s288c["I"][:120_000]), synth(41_234, 5).tobytes(), bytes(s288c["Mito"][:9_000]), synth(3_777, 6).tobytes()]
    tiles = (5120, 7168) if step == 1 else (1024, 2048, 3072, 5120) if step == 5 else (1024, 2048, 3072)
    lags = [2, 3, 5, 11, 50, 99, 101, 200, 255, 511]
    seen = set()
    for tile in tiles:
        w = tile // 256
        for lag in lags + [128 * w - 1, 128 * w]:             # the last two: just inside / outside the baked form
            # (a ctg with fewer windows than the lag is refused like the reference's panic, stat.rs:30)
            seqs = [sq for sq in pool if (len(sq) - 100) // step + 1 >= lag]
            if not seqs:
                continue
            ss = engine.SeqSet(eng, seqs)
            for thr in (3.0,) if lag not in (50, 200) else (3.0, 1.5):
                try:
                    plan = engine.WavePlan(eng, ss, 100, step, lag, thr, 1.0, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE,
                                           tile_windows=tile)
                except _lib.GamsError as e:
                    assert e.code == _lib.EUNSUPPORTED, (step, lag, tile)
                    continue
                seen.add(plan.kernel_name())
                plan.run()
                pk = plan.peaks()
                for c, sq in enumerate(seqs):
                    ocnt, _, osig = ora.wave_windows(sq, 100, step, lag, thr, 1.0)
                    cnt, sig = plan.dense(c)
                    assert np.array_equal(cnt, ocnt) and np.array_equal(sig.astype(np.int32), osig), (step, lag, tile, c)
                    mine = pk[pk["ctg"] == c]
                    idx = np.flatnonzero(osig)
                    assert np.array_equal(mine["window"], idx) and np.array_equal(mine["signal"], osig[idx]), (step, lag, tile, c)
                plan.close()
            ss.close()
    assert any(f", 100, {step}, 0, " in k for k in seen), seen     # the lag-as-argument instantiations did run


@pytest.mark.parametrize("threads", [64, 128, 256])
def test_step1_tiles_of_one_two_or_four_waves(eng, s288c, threads):
    """The step-1 W = 28 kernels run a tile per 64, 128 or 256 threads (gams_wave_plan_set_threads): counts, signals and
    peaks against the oracle on ragged ctgs, lag baked (100) and as an argument, at the lags around the narrow tile's
    layout boundaries (lag + 1 <= threads / 2 * 28, the PS blocks of 28 slots), dense rows and peaks."""
    pool = [bytes(s288c["I"][:90_000]), synth(41_234, 5).tobytes(), bytes(s288c["Mito"][:9_000]), synth(1_777, 6).tobytes(),
            synth(1_691 + 99, 7).tobytes(), synth(1_692 + 99, 8).tobytes(), synth(3_475 + 99, 9).tobytes()]
    half = threads // 2 * 28
    seen = set()
    for lag in [100, 2, 27, 28, 29, 55, 56, 57, 99, 101, 200, half - 2, half - 1, half]:
        seqs = [sq for sq in pool if len(sq) - 99 >= lag]
        ss = engine.SeqSet(eng, seqs)
        for thr in (3.0, 1.5) if lag in (100, 56) else (3.0,):
            plan = engine.WavePlan(eng, ss, 100, 1, lag, thr, 1.0, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE, tile_windows=7168)
            plan.set_threads(threads)
            seen.add(plan.kernel_name())
            plan.run()
            pk = plan.peaks()
            for c, sq in enumerate(seqs):
                ocnt, _, osig = ora.wave_windows(sq, 100, 1, lag, thr, 1.0)
                cnt, sig = plan.dense(c)
                assert np.array_equal(cnt, ocnt) and np.array_equal(sig.astype(np.int32), osig), (threads, lag, thr, c)
                mine = pk[pk["ctg"] == c]
                idx = np.flatnonzero(osig)
                assert np.array_equal(mine["window"], idx) and np.array_equal(mine["signal"], osig[idx]), (threads, lag, c)
            plan.close()
        ss.close()
    tag = f", {threads}"
    assert f"wave_fast_kernel<28, 100, 1, 100, false{tag}>" in seen and f"wave_fast_kernel<28, 100, 1, 0, false{tag}>" in seen, seen


@pytest.mark.parametrize("threads", [64, 128, 256])
def test_step5_tiles_of_one_two_or_four_waves(eng, s288c, threads):
    """The W = 20 step-5 kernel (size and step baked, the lag an argument) in tiles of 64, 128 or 256 threads (a peaks-only
    plan over a genome takes 128): counts, signals, peaks against the oracle on ragged ctgs at the lags around the
    narrow tile's boundaries (lag + 1 <= threads / 2 * 20, the blocks of 20 slots)."""
    pool = [bytes(s288c["I"][:200_000]), synth(81_234, 5).tobytes(), bytes(s288c["Mito"][:30_000]), synth(8_777, 6).tobytes(),
            synth((1_280 - 101) * 5 + 99, 7).tobytes(), synth((1_280 - 100) * 5 + 99, 8).tobytes()]
    half = min(threads // 2 * 20, 656)              # (lag * size beyond 16 bits leaves the fast kernels)
    for lag in [100, 2, 19, 20, 21, 50, 200, 255, half - 2, half - 1]:
        seqs = [sq for sq in pool if (len(sq) - 100) // 5 + 1 >= lag]
        ss = engine.SeqSet(eng, seqs)
        plan = engine.WavePlan(eng, ss, 100, 5, lag, 3.0 if lag != 50 else 1.5, 1.0, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE,
                               tile_windows=5120)
        plan.set_threads(threads)
        assert plan.kernel_name() == f"wave_fast_kernel<20, 100, 5, 0, false, {threads}>", (lag, plan.kernel_name())
        plan.run()
        pk = plan.peaks()
        for c, sq in enumerate(seqs):
            ocnt, _, osig = ora.wave_windows(sq, 100, 5, lag, 3.0 if lag != 50 else 1.5, 1.0)
            cnt, sig = plan.dense(c)
            assert np.array_equal(cnt, ocnt) and np.array_equal(sig.astype(np.int32), osig), (threads, lag, c)
            mine = pk[pk["ctg"] == c]
            idx = np.flatnonzero(osig)
            assert np.array_equal(mine["window"], idx) and np.array_equal(mine["signal"], osig[idx]), (threads, lag, c)
        plan.close()
        ss.close()


@pytest.mark.parametrize("threads", [64, 128])
def test_headline_kernel_in_narrow_workgroups_on_request(eng, s288c, threads):
    """The W = 12 headline kernel also compiles for one or two waves per tile (a diagnostic: it is HBM bound and loses
    3-10 % that way, profiles/r03_barrier_skew.txt): same counts, signals and peaks."""
    seqs = [bytes(s288c["I"]), synth(33_333, 3).tobytes(), bytes(s288c["Mito"][:20_000]), synth(7_680 + 99, 4).tobytes()]
    ss = engine.SeqSet(eng, seqs)
    plan = engine.WavePlan(eng, ss, 100, 10, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE, tile_windows=3072)
    plan.set_threads(threads)
    assert plan.kernel_name() == f"wave_fast_kernel<12, 100, 10, 100, false, {threads}>"
    plan.run()
    pk = plan.peaks()
    for c, sq in enumerate(seqs):
        ocnt, _, osig = ora.wave_windows(sq, 100, 10, 100, 3.0, 1.0)
        cnt, sig = plan.dense(c)
        assert np.array_equal(cnt, ocnt) and np.array_equal(sig.astype(np.int32), osig), (threads, c)
        mine = pk[pk["ctg"] == c]
        idx = np.flatnonzero(osig)
        assert np.array_equal(mine["window"], idx) and np.array_equal(mine["signal"], osig[idx])
    plan.close()
    ss.close()


def test_plans_sharing_a_kernel_function_keep_their_lds(eng, s288c):
    """The dynamic-LDS limit is an attribute of the kernel FUNCTION, not of a plan: plan A (lag 250:
    more K slots) launches, then plan B (lag 20, same instantiation <W,100,10,0>), then A again -- A's second
    launch still has its LDS and the same peaks.  Same for the general tile kernel, whose tiles differ far more
    (size 300 / lag 210 against size 300 / lag 10)."""
    seq = bytes(s288c["I"][:150_000])
    ss = engine.SeqSet(eng, [seq, synth(60_000, 91).tobytes()])
    for big, small in (((100, 10, 250), (100, 10, 20)), ((300, 7, 210), (300, 7, 10))):
        a = engine.WavePlan(eng, ss, big[0], big[1], big[2], 3.0, 1.0, flags=_lib.WAVE_PEAKS)
        b = engine.WavePlan(eng, ss, small[0], small[1], small[2], 3.0, 1.0, flags=_lib.WAVE_PEAKS)
        assert a.kernel_name() == b.kernel_name()
        a.run()
        first = a.peaks().copy()
        b.run()
        pb = b.peaks().copy()
        a.run()
        again = a.peaks()
        assert np.array_equal(first, again)
        for plan, pk, prm in ((a, first, big), (b, pb, small)):
            for c, sq in enumerate((seq, synth(60_000, 91).tobytes())):
                _, _, osig = ora.wave_windows(sq, prm[0], prm[1], prm[2], 3.0, 1.0)
                assert np.array_equal(pk[pk["ctg"] == c]["window"], np.flatnonzero(osig)), (prm, c)
        a.close()
        b.close()
    ss.close()


def test_thresholding_sample_through_the_device(eng):
    """The reference's own unit test of the detector (stat.rs:58-81: 74 values, lag 30, threshold 5, influence 0)
    through the GPU path.  Its values have one decimal and reach 5.0; gc_content cannot exceed 1, so the series is
    scaled by 1/10 -- value y becomes a window of 100 bases with 10*y G's (k as f32 / 100 as f32) -- which leaves
    every comparison of the (scale-free) detector where it was: the expected signals are the reference's vector."""
    data = [1.0, 1.0, 1.1, 1.0, 0.9, 1.0, 1.0, 1.1, 1.0, 0.9, 1.0, 1.1, 1.0, 1.0, 0.9, 1.0, 1.0, 1.1, 1.0, 1.0,
            1.0, 1.0, 1.1, 0.9, 1.0, 1.1, 1.0, 1.0, 0.9, 1.0, 1.1, 1.0, 1.0, 1.1, 1.0, 0.8, 0.9, 1.0, 1.2, 0.9,
            1.0, 1.0, 1.1, 1.2, 1.0, 1.5, 1.0, 3.0, 2.0, 5.0, 3.0, 2.0, 1.0, 1.0, 1.0, 0.9, 1.0, 1.0, 3.0, 2.6,
            4.0, 3.0, 3.2, 2.0, 1.0, 1.0, 0.8, 4.0, 4.0, 2.0, 2.5, 1.0, 1.0, 1.0]
    exp = [0] * 45 + [1, 0, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0]
    assert len(data) == len(exp) == 74
    seq = b"".join(b"G" * round(10 * y) + b"A" * (100 - round(10 * y)) for y in data)
    for step_size, seqs in ((100, [seq]), (100, [seq, synth(30000, 3).tobytes(), seq])):
        ss = engine.SeqSet(eng, seqs)
        plan = engine.WavePlan(eng, ss, 100, step_size, 30, 5.0, 0.0, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE)
        assert plan.kernel_name() == "jac_eval_kernel"
        plan.run()
        pk = plan.peaks()
        for c, sq in enumerate(seqs):
            cnt, sig = plan.dense(c)
            if sq is seq:
                assert cnt.tolist() == [round(10 * y) for y in data]
                assert sig.tolist() == exp
                assert pk[pk["ctg"] == c]["window"].tolist() == [i for i, v in enumerate(exp) if v]
            _, _, osig = ora.wave_windows(sq, 100, step_size, 30, 5.0, 0.0)
            assert np.array_equal(sig.astype(np.int32), osig)
        plan.close()
        ss.close()


@pytest.mark.parametrize("size,step,lag,thr", [(100, 10, 100, 2.0), (100, 10, 100, 1.0), (100, 10, 100, 3.0), (100, 1, 100, 2.0),
                                               (100, 10, 30, 1.5), (100, 5, 200, 2.5), (50, 7, 33, 0.5), (100, 10, 100, 0.05),
                                               (200, 20, 64, 2.0), (100, 10, 2, 2.0), (100, 10, 3, 1.0)])
def test_influence_zero_freezes_and_the_sweeps_follow(eng, s288c, size, step, lag, thr):
    """influence 0: once `lag` windows in a row have signalled filtered[] never moves again and every later window signals
    unless its count equals the frozen one -- the dense regime that plain sweeps cannot settle.  The jac0_* kernels (fill-
    forward filter, the freeze guess from a (size + 1)^2 table) reach the reference's answer WITHOUT the one-wavefront-
    per-ctg recurrence: counts, signals, peaks against the oracle on ragged ctgs (real sequence, synthetic, N runs,
    homopolymer stretches), and the pass reports how it settled."""
    pool = [bytes(s288c["I"][:150_000]), synth(60_000, 41).tobytes(), bytes(s288c["Mito"][:30_000]),
            synth(9_000, 51, gc=0.5, nrate=0.02).tobytes(), (b"ACGT" * 3000 + b"N" * 500 + b"GGCC" * 2000 + b"AT" * 4000)]
    seqs = [sq for sq in pool if (len(sq) - size) // step + 1 >= lag]
    ss = engine.SeqSet(eng, seqs)
    plan = engine.WavePlan(eng, ss, size, step, lag, thr, 0.0, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE)
    frozen = 0
    for rep in range(2):
        plan.run()
        sweeps, serial = plan.settled()
        assert not serial and 0 < sweeps <= 240, (sweeps, serial)
    pk = plan.peaks()
    for c, sq in enumerate(seqs):
        ocnt, _, osig = ora.wave_windows(sq, size, step, lag, thr, 0.0)
        cnt, sig = plan.dense(c)
        assert np.array_equal(cnt, ocnt)
        bad = np.flatnonzero(sig.astype(np.int32) != osig)
        assert bad.size == 0, (c, bad[:5], sig[bad[:5]], osig[bad[:5]])
        mine = pk[pk["ctg"] == c]
        idx = np.flatnonzero(osig)
        assert np.array_equal(mine["window"], idx) and np.array_equal(mine["signal"], osig[idx])
        frozen += int(np.count_nonzero(osig) > osig.size // 2)
    if thr <= 2.0 and lag >= 30:
        assert frozen > 0                      # (the case this test is about did occur)
    plan.close()
    ss.close()


@pytest.mark.parametrize("seed", range(6))
def test_influence_fuzz_against_the_oracle(eng, s288c, seed):
    """Random (size, step, lag, threshold, influence) with influence != 1 on ragged batches: thresholds low enough
    that most windows signal (many sweeps; at influence 0 runs of thousands of signalled windows, which hand the batch
    to the one-wavefront-per-ctg recurrence), lags far beyond a tile, sizes beyond 255 (generic tile kernel), steps beyond
    a tile (untiled kernels)."""
    rng = np.random.default_rng(1000 + seed)
    pool = [bytes(s288c["I"][:90_000]), synth(33_333, 40 + seed).tobytes(), bytes(s288c["Mito"][:20_000]),
            synth(8_000, 50 + seed, gc=0.5, nrate=0.02).tobytes(), (b"ACGT" * 3000 + b"N" * 500 + b"GGCC" * 2000)]
    cases = []
    for _ in range(7):
        size = int(rng.choice([20, 50, 100, 100, 100, 256, 300]))
        step = int(rng.choice([1, 3, 10, 10, 25, 50, 2000]))
        lag = int(rng.choice([2, 3, 5, 30, 64, 100, 100, 200, 401]))
        thr = float(rng.choice([0.3, 1.0, 2.0, 3.0, 3.0, 5.0]))
        infl = float(rng.choice([0.0, 0.0, 0.25, 0.5, 0.9, 0.999, 1.5, -0.5]))
        cases.append((size, step, lag, thr, infl))
    cases += [(100, 10, 598, 2.0, 0.5), (100, 10, 1999, 2.0, 0.5)] if seed == 0 else []
    cases += [(100, 10, 100, 0.05, 0.0), (100, 1, 50, 1.0, 0.0), (100, 10, 100, 2.0, 0.0), (100, 10, 100, 1.0, 0.999)] if seed == 1 else []
    for size, step, lag, thr, infl in cases:
        seqs = [sq for sq in pool if (len(sq) - size) // step + 1 >= lag]
        if not seqs:
            continue
        ss = engine.SeqSet(eng, seqs)
        plan = engine.WavePlan(eng, ss, size, step, lag, thr, infl, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE)
        for rep in range(2):                   # twice: the second pass runs over the first one's tables
            plan.run()
        pk = plan.peaks()
        for c, sq in enumerate(seqs):
            ocnt, _, osig = ora.wave_windows(sq, size, step, lag, thr, infl)
            cnt, sig = plan.dense(c)
            assert np.array_equal(cnt, ocnt), (size, step, lag, thr, infl, c)
            bad = np.flatnonzero(sig.astype(np.int32) != osig)
            assert bad.size == 0, (size, step, lag, thr, infl, c, bad[:5], sig[bad[:5]], osig[bad[:5]])
            idx = np.flatnonzero(osig)
            mine = pk[pk["ctg"] == c]
            assert np.array_equal(mine["window"], idx) and np.array_equal(mine["signal"], osig[idx])
        plan.close()
        ss.close()
