"""Parity at BASELINE.json's full sizes.

configs[1] (S288c, 12 Mb) and configs[2] (A. thaliana, 120 Mb) are small enough for the CPU
oracle to process completely (seconds), so the GPU peaks are compared record by record.
configs[3] geometry (step 1) is checked on one 60-Mb chromosome the same way, plus the
size-independent properties: identical results for every tiling / kernel variant, and for a
ctg processed alone versus inside a batch."""
import numpy as np
import pytest

from gams_amd import _lib, engine, synth
from oracle import oracle as ora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = engine.Engine(0)
    yield e
    e.close()


def oracle_peaks(ctgs, size, step, lag, thr):
    out = []
    for c, ctg in enumerate(ctgs):
        cnt, _, sig = ora.wave_windows(ctg["seq"], size, step, lag, thr, 1.0)
        idx = np.flatnonzero(sig)
        rec = np.zeros(idx.size, _lib.PEAK_DTYPE)
        rec["ctg"], rec["window"], rec["gc_count"], rec["signal"] = c, idx, cnt[idx], sig[idx]
        out.append(rec)
    return np.concatenate(out) if out else np.zeros(0, _lib.PEAK_DTYPE)


def gpu_peaks(eng, ss, size, step, lag, thr, tile=0):
    plan = engine.WavePlan(eng, ss, size, step, lag, thr, 1.0, flags=_lib.WAVE_PEAKS, tile_windows=tile)
    plan.run()
    pk = plan.peaks()
    n = plan.total_windows
    plan.close()
    return pk, n


def test_config1_s288c_full_genome(eng):
    ctgs = synth.genome_ctgs(synth.S288C_LENGTHS, 500000)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    exp = oracle_peaks(ctgs, 100, 10, 100, 3.0)
    for tile in (0, 1024, 2048, 3072, 5120, 1536):  # W = 4 / 8 / 12 / 20; 1536: the general (prefix-array) kernel
        pk, n = gpu_peaks(eng, ss, 100, 10, 100, 3.0, tile)
        assert n == 1215506
        assert np.array_equal(pk, exp), tile
    # four passes in flight (the bench's mode; the default tile becomes W = 8 here): every held pass
    plan = engine.WavePlan(eng, ss, 100, 10, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    plan.set_depth(4)
    plan.run_n(103)
    for age in range(4):
        plan.select(age)
        assert np.array_equal(plan.peaks(), exp), age
    plan.close()
    ss.close()


def test_config2_atha_full_genome(eng):
    ctgs = synth.genome_ctgs(synth.ATHA_LENGTHS, 500000)
    assert sum(len(c["seq"]) for c in ctgs) > 119_000_000
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    pk, n = gpu_peaks(eng, ss, 100, 10, 100, 3.0)
    exp = oracle_peaks(ctgs, 100, 10, 100, 3.0)
    assert np.array_equal(pk, exp)
    assert 0.002 < pk.size / n < 0.05              # peak density like results/Atha.md:217-221
    # a ctg alone gives the same records as inside the batch
    k = len(ctgs) // 2
    s1 = engine.SeqSet(eng, [ctgs[k]["seq"]])
    one, _ = gpu_peaks(eng, s1, 100, 10, 100, 3.0)
    ref = pk[pk["ctg"] == k].copy()
    ref["ctg"] = 0
    assert np.array_equal(one, ref)
    s1.close()
    ss.close()


def test_config3_step1_geometry_one_chromosome(eng):
    chrom = synth.chromosome(60_000_000, 77)
    ctgs = synth.gen_ctgs("77", chrom, piece=1000000)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    pk, n = gpu_peaks(eng, ss, 100, 1, 100, 3.0)
    assert n > 59_000_000
    exp = oracle_peaks(ctgs, 100, 1, 100, 3.0)
    assert np.array_equal(pk, exp)
    for tile in (1024, 5120, 4096):
        other, _ = gpu_peaks(eng, ss, 100, 1, 100, 3.0, tile)
        assert np.array_equal(other, pk), tile
    ss.close()


def test_guard_band_margin_on_the_atha_genome(eng):
    """on the configs[2] genome the default guard band (safety 1.5), the bare bound
    (safety 1.0) and the all-exact run give identical peaks (= the oracle's, checked above), and the
    default run sends fewer than 1e-3 of the windows down the exact path (stat.rs:36-38)."""
    ctgs = synth.genome_ctgs(synth.ATHA_LENGTHS, 500000)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    plan = engine.WavePlan(eng, ss, 100, 10, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    plan.run()
    ref = plan.peaks()
    n = plan.total_windows
    n_default = plan.exact_count()
    assert 0 < n_default < 1e-3 * n
    plan.set_guard(1.0, False)
    plan.run()
    assert np.array_equal(plan.peaks(), ref)
    assert plan.exact_count() <= n_default
    plan.close()
    # all-exact at full size is minutes of one-window-per-wave work: the largest chromosome's first 40 ctgs
    sub = ctgs[:40]
    s2 = engine.SeqSet(eng, [c["seq"] for c in sub])
    p2 = engine.WavePlan(eng, s2, 100, 10, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    p2.set_guard(1.5, True)
    p2.run()
    got = p2.peaks()
    assert np.array_equal(got, ref[ref["ctg"] < 40])
    assert 0.99 * (p2.total_windows - 100 * len(sub)) <= p2.exact_count() <= p2.total_windows - 100 * len(sub)
    p2.close()
    s2.close()
    ss.close()
