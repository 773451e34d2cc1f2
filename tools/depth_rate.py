#!/usr/bin/env python3
"""Passes per second of one plan at depth 1..4, queued from Python call by call and from C (run_n)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

eng = engine.Engine(0)
ctgs = synth.genome_ctgs(synth.S288C_LENGTHS, 500000)
ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
plan = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS)
nw = plan.total_windows
K = 1000
for depth in (1, 2, 3, 4):
    plan.set_depth(depth)
    plan.run_n(20)
    eng.sync()
    res = []
    for mode in ("python loop", "run_n"):
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            if mode == "run_n":
                plan.run_n(K)
            else:
                for _ in range(K):
                    plan.run()
            tq = time.perf_counter() - t0
            eng.sync()
            best = min(best, (time.perf_counter() - t0) / K)
        res.append(f"{mode}: {best * 1e6:.2f} us/pass ({nw / best / 1e9:.0f} G windows/s, queueing {tq / K * 1e6:.2f} us/pass)")
    print(f"depth {depth}: " + "; ".join(res), flush=True)
