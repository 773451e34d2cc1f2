"""Randomised parameter sweep of the wave path on the GPU against the oracle: sizes, steps
(smaller than, equal to and larger than the size), lags and thresholds drawn at random, ragged
ctg lengths, every kernel variant (baked / run-time fast kernels, general kernel, serial path)."""
import os

import numpy as np
import pytest

from gams_amd import _lib, engine
from oracle import oracle as ora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = engine.Engine(0)
    yield e
    e.close()


def random_seq(rng, n):
    # blocks of different GC richness, soft-masked stretches, N runs, a homopolymer
    s = np.empty(n, np.uint8)
    pos = 0
    while pos < n:
        ln = int(rng.integers(200, 6000))
        gc = rng.uniform(0.15, 0.75)
        blk = np.where(rng.random(ln) < gc, rng.choice([0x47, 0x43], ln), rng.choice([0x41, 0x54], ln)).astype(np.uint8)
        if rng.random() < 0.3:
            blk |= 0x20
        if rng.random() < 0.1:
            blk[:] = rng.choice([0x4E, 0x41, 0x67])
        s[pos:pos + ln] = blk[:n - pos]
        pos += ln
    return s


# GAMS_FUZZ_SEEDS=first:count widens the sweep (profiles/r01_fuzz_log.txt: 122,000 seeds)
_FIRST, _COUNT = (int(x) for x in os.environ.get("GAMS_FUZZ_SEEDS", "0:64").split(":"))
_SIZE100 = os.environ.get("GAMS_FUZZ_SIZE100", "") == "1"   # size 100, step 1 / 5 / 10 / 20, any lag in 2..599


@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + _COUNT))
def test_random_parameters(eng, seed):
    rng = np.random.default_rng(1000 + seed)
    size = int(rng.choice([1, 7, 10, 50, 64, 100, 100, 100, 128, 200, 255, 256, 300, 1000]))
    step = int(rng.choice([1, 2, 5, 10, 10, 10, 16, 31, 32, 33, 50, 64, 100, 150]))
    lag = int(rng.choice([2, 3, 10, 33, 50, 100, 100, 127, 128, 200, 400]))
    thr = float(rng.choice([0.5, 1.0, 2.0, 2.5, 3.0, 3.0, 3.5, 5.0]))
    infl = float(rng.choice([1.0, 1.0, 1.0, 1.0, 0.5, 0.0]))
    if _SIZE100:                                   # the kernels with size and step baked and the lag as an argument
        size, step, lag = 100, int(rng.choice([1, 5, 10, 20])), int(rng.integers(2, 600))
    n_ctg = int(rng.integers(1, 5))
    need = size + (lag + 5) * step
    seqs = [random_seq(rng, int(need + rng.integers(0, 40 * need // 10 + 5000))) for _ in range(n_ctg)]
    ss = engine.SeqSet(eng, seqs)
    # every parameter set runs: halos beyond a 64-KB tile take the untiled kernels
    plan = engine.WavePlan(eng, ss, size, step, lag, thr, infl, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE)
    plan.run()
    pk = plan.peaks()
    exp = []
    for c, s in enumerate(seqs):
        ocnt, _, osig = ora.wave_windows(s, size, step, lag, thr, infl)
        cnt, sig = plan.dense(c)
        assert np.array_equal(cnt, ocnt), (size, step, lag, thr, infl)
        bad = np.flatnonzero(sig.astype(np.int32) != osig)
        assert bad.size == 0, (size, step, lag, thr, infl, bad[:5])
        idx = np.flatnonzero(osig)
        exp += [(c, int(i), int(ocnt[i]), int(osig[i])) for i in idx]
    got = [(int(r["ctg"]), int(r["window"]), int(r["gc_count"]), int(r["signal"])) for r in pk]
    assert got == exp, (size, step, lag, thr, infl)
    plan.close()
    ss.close()


# GAMS_FUZZ_NARROW=first:count widens this one (profiles/r03_fuzz_log.txt)
_NFIRST, _NCOUNT = (int(x) for x in os.environ.get("GAMS_FUZZ_NARROW", "0:12").split(":"))


@pytest.mark.parametrize("seed", range(_NFIRST, _NFIRST + _NCOUNT))
def test_random_step1_tiles_per_wave(eng, seed):
    """size 100 / step 1 in workgroups of one or two waves (what a peaks-only plan over a genome runs by default): random
    lags up to the narrow tile's limit, thresholds, ragged ctgs with N runs and homopolymers; peaks alone and peaks +
    dense rows, against the oracle."""
    rng = np.random.default_rng(77000 + seed)
    threads = int(rng.choice([64, 64, 128]))
    lag = int(rng.choice([100, 100, int(rng.integers(2, threads // 2 * 28))]))
    thr = float(rng.choice([1.0, 2.0, 3.0, 3.0, 3.5]))
    dense = bool(rng.random() < 0.5)
    n_ctg = int(rng.integers(1, 6))
    seqs = [random_seq(rng, int(99 + lag + 5 + rng.integers(0, 30000))) for _ in range(n_ctg)]
    ss = engine.SeqSet(eng, seqs)
    plan = engine.WavePlan(eng, ss, 100, 1, lag, thr, 1.0, flags=_lib.WAVE_PEAKS | (_lib.WAVE_DENSE if dense else 0),
                           tile_windows=7168)
    plan.set_threads(threads)
    name = plan.kernel_name()                 # (lag * size beyond 16 bits: the general kernel, no narrow form)
    assert name.endswith(f", {threads}>") or not name.startswith("wave_fast_kernel<28"), name
    plan.run()
    pk = plan.peaks()
    exp = []
    for c, s in enumerate(seqs):
        ocnt, _, osig = ora.wave_windows(s, 100, 1, lag, thr, 1.0)
        if dense:
            cnt, sig = plan.dense(c)
            assert np.array_equal(cnt, ocnt) and np.array_equal(sig.astype(np.int32), osig), (threads, lag, thr, c)
        idx = np.flatnonzero(osig)
        exp += [(c, int(i), int(ocnt[i]), int(osig[i])) for i in idx]
    got = [(int(r["ctg"]), int(r["window"]), int(r["gc_count"]), int(r["signal"])) for r in pk]
    assert got == exp, (threads, lag, thr)
    plan.close()
    ss.close()
