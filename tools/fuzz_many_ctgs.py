#!/usr/bin/env python3
"""Batches of many ragged ctgs (tile table, ctg boundaries, halos at ctg starts, depth rotation) against the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gams_amd import _lib, engine  # noqa: E402
from oracle import oracle as ora  # noqa: E402
from test_gpu_random_params import random_seq  # noqa: E402

eng = engine.Engine(0)
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tot_ctgs = tot_win = 0
for it in range(n_iter):
    rng = np.random.default_rng(31337 + it)
    size, step, lag, thr = [(100, 10, 100, 3.0), (100, 1, 100, 3.0), (50, 5, 20, 2.0), (128, 16, 64, 2.5)][it % 4]
    n_ctg = int(rng.integers(20, 300))
    need = size + lag * step
    seqs = [random_seq(rng, int(need + rng.choice([0, 1, 7, 100, 1023, 1024, 1025, 5000, 40000]))) for _ in range(n_ctg)]
    ss = engine.SeqSet(eng, seqs)
    plan = engine.WavePlan(eng, ss, size, step, lag, thr, 1.0, flags=_lib.WAVE_PEAKS)
    depth = int(rng.integers(1, 5))
    plan.set_depth(depth)
    plan.run_n(depth + int(rng.integers(0, 20)))
    exp = []
    for c, s in enumerate(seqs):
        ocnt, _, osig = ora.wave_windows(s, size, step, lag, thr, 1.0)
        idx = np.flatnonzero(osig)
        rec = np.zeros(idx.size, _lib.PEAK_DTYPE)
        rec["ctg"], rec["window"], rec["gc_count"], rec["signal"] = c, idx, ocnt[idx], osig[idx]
        exp.append(rec)
    exp = np.concatenate(exp)
    for age in range(depth):
        plan.select(age)
        pk = plan.peaks()
        assert np.array_equal(pk, exp), (it, age, n_ctg, depth)
    tot_ctgs += n_ctg
    tot_win += plan.total_windows
    plan.close()
    ss.close()
    print(f"iteration {it}: {n_ctg} ctgs, depth {depth}, {exp.size} peaks ok", flush=True)
print(f"many-ctg fuzz: {n_iter} batches, {tot_ctgs} ctgs, {tot_win} windows, every held pass identical to the oracle")
