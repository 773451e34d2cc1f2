#!/usr/bin/env python3
"""End-to-end `sw` (host buffers in -> TSV rows out) on an Atha-chr1-shaped chromosome with 1e5 point features
(the reference: fsw of the T-DNA features takes 14.7 s serial / 1.9 s with 16 threads, doc/benchmark/Atha.md:348,353)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import engine, host, synth  # noqa: E402

eng = engine.Engine(0)
rng = np.random.default_rng(5)
chrom = synth.chromosome(30_000_000, 9)
ctgs = synth.gen_ctgs("9", chrom, piece=1000000)
feats = {}
for c in ctgs:
    nf = 100000 // len(ctgs)
    fs = np.sort(rng.integers(c["chr_start"], c["chr_end"] + 1, nf))
    feats[c["id"]] = [(f"feature:{c['id']}:{i + 1}", int(s), int(s)) for i, s in enumerate(fs)]
for rep in range(3):
    t0 = time.perf_counter()
    rows = 0
    nbytes = 0
    for c in ctgs:
        out = host.sw(eng, c, feats[c["id"]])
        rows += out.count("\n")
        nbytes += len(out)
    dt = time.perf_counter() - t0
    print(f"sw end to end: {rows} rows ({nbytes / 1e6:.0f} MB of text) for {sum(len(v) for v in feats.values())} features "
          f"on {len(ctgs)} ctgs in {dt * 1e3:.0f} ms -> {rows / dt / 1e6:.2f} M rows/s", flush=True)

# the same through the batched host path (sw_proc_ctgs: one seqset, one gams_gpu_sw_batch call, rows read back into
# page-locked memory, ctgs formatted on host threads)
flist = [feats[c["id"]] for c in ctgs]
for rep in range(3):
    t0 = time.perf_counter()
    out = host.sw_multi([eng], ctgs, flist)
    dt = time.perf_counter() - t0
    rows = out.count("\n")
    print(f"sw end to end, batched: {rows} rows ({len(out) / 1e6:.0f} MB of text) in {dt * 1e3:.0f} ms -> "
          f"{rows / dt / 1e6:.2f} M rows/s", flush=True)

# the operator's own time (upload + kernels + row text), without this binding's feature parsing and string copies
for rep in range(3):
    out, ms = host.sw_multi_timed([eng], ctgs, flist)
    rows = out.count("\n")
    print(f"sw operator time: {rows} rows ({len(out) / 1e6:.0f} MB of text) in {ms:.1f} ms -> {rows / ms / 1e3:.2f} M rows/s",
          flush=True)
