#!/usr/bin/env python3
"""influence 0 against the oracle: random sizes, steps, lags and (low) thresholds on ragged batches of real-looking and
degenerate ctgs; counts how the passes settled (sweeps; hand-overs to the one-wavefront-per-ctg recurrence).
usage: tools/fuzz_influence0.py [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402
from oracle import oracle as ora  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
eng = engine.Engine(0)
rng = np.random.default_rng(424242)
chrom = synth.chromosome(6_000_000, 3)
sweeps_seen, serial, windows = [], 0, 0
for case in range(n_cases):
    size = int(rng.choice([20, 50, 100, 100, 100, 200, 400]))
    step = int(rng.choice([1, 2, 5, 10, 10, 10, 20, 50]))
    lag = int(rng.choice([2, 5, 30, 64, 100, 100, 100, 200, 300]))
    thr = float(rng.choice([0.05, 0.5, 1.0, 1.5, 2.0, 2.0, 2.5, 3.0]))
    seqs = []
    for _ in range(int(rng.integers(1, 5))):
        n = int(size + (lag + 3) * step + rng.integers(0, 400_000 if step >= 5 else 60_000))
        off = int(rng.integers(0, chrom.size - n))
        sq = chrom[off:off + n].copy()
        kind = rng.random()
        if kind < 0.15:
            sq[n // 3:n // 3 + int(rng.integers(1, 5000))] = ord("N")
        elif kind < 0.25:
            sq[n // 2:] = np.frombuffer((b"GC" * n)[:n - n // 2], np.uint8)
        seqs.append(sq.tobytes())
    ss = engine.SeqSet(eng, seqs)
    plan = engine.WavePlan(eng, ss, size, step, lag, thr, 0.0, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE)
    plan.run()
    sw, ser = plan.settled()
    sweeps_seen.append(sw)
    serial += int(ser)
    pk = plan.peaks()
    for c, sq in enumerate(seqs):
        ocnt, _, osig = ora.wave_windows(sq, size, step, lag, thr, 0.0)
        cnt, sig = plan.dense(c)
        assert np.array_equal(cnt, ocnt), (size, step, lag, thr, c)
        bad = np.flatnonzero(sig.astype(np.int32) != osig)
        assert bad.size == 0, (size, step, lag, thr, c, bad[:5])
        mine = pk[pk["ctg"] == c]
        idx = np.flatnonzero(osig)
        assert np.array_equal(mine["window"], idx) and np.array_equal(mine["signal"], osig[idx]), (size, step, lag, thr, c)
        windows += int(ocnt.size)
    plan.close()
    ss.close()
print(f"influence-0 fuzz: {n_cases} random configurations, {windows} windows, all identical to the oracle; sweeps queued "
      f"min/median/max {min(sweeps_seen)}/{int(np.median(sweeps_seen))}/{max(sweeps_seen)}, {serial} passes handed to the serial recurrence")
