#!/usr/bin/env python3
"""End-to-end text paths of `locate`, `locate --count` and `anno` (range strings in -> TSV text out) through
the host layer: 1e6 lines on an Atha-shaped ctg table.  The reference: anno of ~1e5-1e6 rg lines 1.8-2.1 s
(doc/benchmark/Atha.md:658,660); locate -f of a T-DNA file 20-48 ms (:486-510)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import engine, host, synth  # noqa: E402

eng = engine.Engine(0)
rng = np.random.default_rng(3)
ctgs = []
for k, ln in enumerate(synth.ATHA_LENGTHS):
    pos = 1
    i = 0
    while pos <= ln:
        end = min(ln, pos + 499999)
        if ln - end < 5000:
            end = ln
        i += 1
        ctgs.append(dict(id=f"ctg:{k + 1}:{i}", chr_id=str(k + 1), chr_start=pos, chr_end=end, seq=b""))
        pos = end + 1
N = 1_000_000
pick = rng.integers(0, len(ctgs), N)
starts = np.array([c["chr_start"] for c in ctgs])[pick] + rng.integers(0, 400000, N)
ends = starts + rng.integers(0, 2000, N)
rgs = [f"{ctgs[p]['chr_id']}:{s}-{e}" for p, s, e in zip(pick, starts, ends)]
t0 = time.perf_counter()
out = host.locate(eng, ctgs, rgs)
t1 = time.perf_counter()
print(f"locate: {N} ranges -> {out.count(chr(10))} lines in {(t1 - t0) * 1e3:.0f} ms (the operator under the binding: {host.last_operator_ms():.1f} ms)")
recs = [(ctgs[p]["id"], r) for p, r in zip(pick[:300000], rgs[:300000])]
t0 = time.perf_counter()
out = host.locate(eng, ctgs, rgs, count=True, rg_records=recs)
t1 = time.perf_counter()
print(f"locate --count: {N} ranges against {len(recs)} stored rg -> {out.count(chr(10))} lines in {(t1 - t0) * 1e3:.0f} ms (operator: {host.last_operator_ms():.1f} ms)")
runlists = {}
for k, ln in enumerate(synth.ATHA_LENGTHS):
    cuts = np.sort(rng.choice(np.arange(1, ln, 7), min(60000, ln // 28 * 2), replace=False))
    runlists[str(k + 1)] = ",".join(f"{a}-{b - 1}" for a, b in zip(cuts[0::2], cuts[1::2]))
lines = [f"rg:{ctgs[p]['id']}:{i}\t{r}" for i, (p, r) in enumerate(zip(pick, rgs))]
t0 = time.perf_counter()
out = host.anno(eng, ctgs, runlists, lines, header=False, idx_id=1, idx_range=2)
t1 = time.perf_counter()
print(f"anno: {N} lines against {sum(v.count(',') + 1 for v in runlists.values())} spans -> {out.count(chr(10))} lines in {(t1 - t0) * 1e3:.0f} ms (operator: {host.last_operator_ms():.1f} ms)")
