#!/usr/bin/env python3
"""`gams wave` then `gams peak` (peak.rs:24-177) on an A. thaliana-shaped genome through the host layer: the wave TSV rows
(65 k) in -> the Peak records out."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import engine, host, synth  # noqa: E402

eng = engine.Engine(0)
ctgs = [dict(id=c["id"], chr_id=c["chr_id"], chr_start=c["chr_start"], chr_end=c["chr_end"], seq=c["seq"])
        for c in synth.genome_ctgs(synth.ATHA_LENGTHS, 500000)]
rows = host.wave(eng, ctgs).splitlines()
print(len(rows), "wave rows")
for rep in range(3):
    t0 = time.perf_counter()
    out = host.peak(eng, ctgs, rows)
    dt = time.perf_counter() - t0
    print(f"peak: {out.count(chr(10))} records in {dt * 1e3:.0f} ms", flush=True)
