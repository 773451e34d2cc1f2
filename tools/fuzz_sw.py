#!/usr/bin/env python3
"""Random sweep of `sw` (sizes, max, resize, ctg placement, feature lengths, ctg edges) against the oracle:
every text field of every row must be identical."""
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
rows_total = 0
for it in range(n_iter):
    if it % 100 == 0:
        print('iteration', it, flush=True)
    rng = np.random.default_rng(777 + it)
    chrom = s288c["I"] if rng.random() < 0.7 else s288c["Mito"]
    ln = int(rng.integers(5000, min(len(chrom), 120000)))
    off = int(rng.integers(0, len(chrom) - ln + 1))
    c = dict(id="ctg:X:1", chr_id="X", chr_start=off + 1, chr_end=off + ln, seq=bytes(chrom[off:off + ln]))
    size = int(rng.choice([2, 3, 7, 10, 33, 50, 100, 100, 128, 200, 255, 256, 1000]))
    mx = int(rng.choice([0, 1, 2, 5, 20, 20, 40]))
    resize = int(rng.choice([2, 3, size, size + 1, 99, 100, 333, 500, 500, 5000]))
    feats = []
    for i in range(int(rng.integers(1, 120))):
        s = int(rng.integers(c["chr_start"], c["chr_end"] + 1))
        e = min(c["chr_end"], s + int(rng.choice([0, 0, 1, 2, 3, 99, 100, 101, 1500, 20000])))
        if rng.random() < 0.2:                       # at or near an edge of the ctg
            s = int(rng.choice([c["chr_start"], c["chr_start"] + 1, c["chr_end"] - 1, c["chr_end"]]))
            e = min(c["chr_end"], s + int(rng.choice([0, 1, 50])))
        feats.append((f"feature:{c['id']}:{i + 1}", s, e))
    got = host.sw(eng, c, feats, size, mx, resize)
    exp = ora.sw_proc_ctg(c["chr_id"], c["chr_start"], c["chr_end"], c["seq"], feats, size, mx, resize)
    if got != exp:
        g, x = got.splitlines(), exp.splitlines()
        bad = [i for i, (a, b) in enumerate(zip(g, x)) if a != b][:3]
        print("MISMATCH", it, (size, mx, resize), len(g), len(x), [(g[i], x[i]) for i in bad])
        sys.exit(1)
    rows_total += got.count("\n")
print(f"sw fuzz: {n_iter} random configurations, {rows_total} rows, all identical to the oracle")
