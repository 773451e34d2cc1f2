#!/usr/bin/env python3
"""End-to-end rate of the host operator (host buffers in -> TSV rows out): upload over PCIe,
kernel, peak readback, merge + formatting.  Never the headline `value` (that one has the inputs
resident in HBM); reported in DESIGN.md."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import engine, host, synth  # noqa: E402

eng = engine.Engine(0)
ctgs = [dict(id=c["id"], chr_id=c["chr_id"], chr_start=c["chr_start"], chr_end=c["chr_end"], seq=c["seq"])
        for c in synth.genome_ctgs(synth.SYNTH384_LENGTHS, 1000000, first_chr_index=500)]
bases = sum(len(c["seq"]) for c in ctgs)
windows = sum((len(c["seq"]) - 100) // 10 + 1 for c in ctgs)
host.wave(eng, ctgs[:4])   # warm up (library load, first allocations)
for label, fn in (("one batch", lambda: host.wave(eng, ctgs)),
                  ("pipelined 64-MB batches", lambda: host.wave_multi([eng], ctgs, batch_bytes=64 << 20)),
                  ("pipelined 16-MB batches", lambda: host.wave_multi([eng], ctgs, batch_bytes=16 << 20))):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        out = fn()
        best = min(best, time.perf_counter() - t0)
    print(f"{label}: {best * 1e3:.1f} ms for {bases} bases, {windows} windows, {out.count(chr(10))} rows -> "
          f"{windows / best / 1e9:.2f} G windows/s, {bases / best / 1e9:.2f} GB/s of sequence")

small = [dict(id=c["id"], chr_id=c["chr_id"], chr_start=c["chr_start"], chr_end=c["chr_end"], seq=c["seq"])
         for c in synth.genome_ctgs(synth.S288C_LENGTHS, 500000)]
sb = sum(len(c["seq"]) for c in small)
sw_ = sum((len(c["seq"]) - 100) // 10 + 1 for c in small)
best = 1e9
for _ in range(10):
    t0 = time.perf_counter()
    out = host.wave(eng, small)
    best = min(best, time.perf_counter() - t0)
print(f"S288c one batch: {best * 1e3:.2f} ms for {sb} bases, {sw_} windows, {out.count(chr(10))} rows -> "
      f"{sw_ / best / 1e9:.2f} G windows/s, {sb / best / 1e9:.2f} GB/s of sequence")

# the reference's own published case: A. thaliana, wave size 100 step 10 lag 100, 3.169 s wall with -p 8
# (results/Atha.md:207-215, incl. Redis GET + gunzip); here host buffers in -> TSV rows out
atha = [dict(id=c["id"], chr_id=c["chr_id"], chr_start=c["chr_start"], chr_end=c["chr_end"], seq=c["seq"])
        for c in synth.genome_ctgs(synth.ATHA_LENGTHS, 500000)]
ab = sum(len(c["seq"]) for c in atha)
aw = sum((len(c["seq"]) - 100) // 10 + 1 for c in atha)
best = 1e9
for _ in range(4):
    t0 = time.perf_counter()
    out = host.wave(eng, atha)
    best = min(best, time.perf_counter() - t0)
print(f"Atha-shaped one batch: {best * 1e3:.2f} ms for {ab} bases, {aw} windows, {out.count(chr(10))} rows -> "
      f"{aw / best / 1e9:.2f} G windows/s, {ab / best / 1e9:.2f} GB/s of sequence")
