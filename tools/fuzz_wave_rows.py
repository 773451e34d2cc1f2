#!/usr/bin/env python3
"""Random sweep of the `gams wave` TSV rows (peak collection, merge_ints with random coverage, range
and float text, --signal mode) produced by the host operator against the oracle's proc_ctg."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from gams_amd import engine, host  # noqa: E402
from oracle import oracle as ora  # noqa: E402

eng = engine.Engine(0)
s288c = helpers.load_s288c()
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rows_total = merged_total = skipped = 0
for it in range(n_iter):
    if it % 50 == 0:
        print("iteration", it, flush=True)
    rng = np.random.default_rng(4242 + it)
    size = int(rng.choice([10, 50, 100, 100, 100, 128, 200, 255, 300]))
    step = int(rng.choice([1, 2, 5, 10, 10, 10, 25, 50, 100, 150]))
    lag = int(rng.choice([5, 20, 50, 100, 100, 200]))
    thr = float(rng.choice([1.0, 2.0, 2.5, 3.0, 3.0, 3.5]))
    infl = float(rng.choice([1.0, 1.0, 1.0, 0.5, 0.0]))
    cov = float(rng.choice([0.05, 0.1, 0.2, 0.2, 0.5, 0.9, 1.0]))
    sig = bool(rng.random() < 0.15)
    ctgs = []
    for k in range(int(rng.integers(1, 4))):
        chrom = s288c["I"] if rng.random() < 0.7 else s288c["Mito"]
        need = size + (lag + 5) * step
        ln = int(rng.integers(need, min(len(chrom), need + 40000)))
        off = int(rng.integers(0, len(chrom) - ln + 1))
        ctgs.append(dict(id=f"ctg:X:{k + 1}", chr_id="X", chr_start=off + 1, chr_end=off + ln,
                         seq=bytes(chrom[off:off + ln])))
    got = host.wave(eng, ctgs, size, step, lag, thr, infl, cov, sig)
    skipped += (lag + 1) * step + size + 256 * step > 65520
    exp = "".join(ora.wave_proc_ctg(c["chr_id"], c["chr_start"], c["chr_end"], c["seq"], size, step, lag, thr, infl,
                                    cov, sig) for c in ctgs)
    if got != exp:
        g, x = got.splitlines(), exp.splitlines()
        bad = [i for i, (a, b) in enumerate(zip(g, x)) if a != b][:3]
        print("MISMATCH", it, (size, step, lag, thr, infl, cov, sig), len(g), len(x), [(g[i], x[i]) for i in bad])
        sys.exit(1)
    rows_total += got.count("\n")
    merged_total += got.count("(+):")
print(f"wave rows fuzz: {n_iter} random configurations, {rows_total} rows ({merged_total} merged ranges), "
      f"all identical to the oracle; {skipped} of them through the untiled kernels (halo beyond the 64-KB tile)")
