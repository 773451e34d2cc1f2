#!/usr/bin/env python3
"""`gams wave --signal` end to end (a row for EVERY window, wave.rs:158-168) on an A. thaliana-shaped genome: host buffers
in -> TSV text out through the host layer."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import engine, host, synth  # noqa: E402

eng = engine.Engine(0)
ctgs = [dict(id=c["id"], chr_id=c["chr_id"], chr_start=c["chr_start"], chr_end=c["chr_end"], seq=c["seq"])
        for c in synth.genome_ctgs(synth.ATHA_LENGTHS, 500000)]
host.wave(eng, ctgs[:2], is_signal=True)
for rep in range(3):
    t0 = time.perf_counter()
    text, st = host.wave_timed(eng, ctgs, is_signal=True)
    dt = time.perf_counter() - t0
    print(f"--signal, 120 Mb: {dt * 1e3:.0f} ms through the binding, the operator {st['total_ms']:.1f} ms (upload {st['upload_ms']:.1f}, "
          f"kernel {st['kernel_ms']:.2f}, rows {st['peaks_ms']:.1f} + per-ctg strings {st['format_ms']:.1f}); "
          f"{text.count(bytes([10]))} rows, {len(text) / 1e6:.0f} MB of text", flush=True)
