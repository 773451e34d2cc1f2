#!/usr/bin/env python3
"""End to end at step 1 (the reference's O(p^2) merge takes minutes per genome there):
host buffers in -> TSV rows out for a 384-Mb genome, size 100 / step 1 / lag 100."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import engine, host, synth  # noqa: E402

eng = engine.Engine(0)
ctgs = [dict(id=c["id"], chr_id=c["chr_id"], chr_start=c["chr_start"], chr_end=c["chr_end"], seq=c["seq"])
        for c in synth.genome_ctgs(synth.SYNTH384_LENGTHS, 1000000, first_chr_index=500)]
host.wave(eng, ctgs[:2], step=1)
for rep in range(2):
    t0 = time.perf_counter()
    out = host.wave(eng, ctgs, step=1)
    dt = time.perf_counter() - t0
    print(f"step 1, 384 Mb: {dt * 1e3:.0f} ms, {out.count(chr(10))} rows ({out.count('(+):')} merged), {len(out) / 1e6:.0f} MB of text", flush=True)
