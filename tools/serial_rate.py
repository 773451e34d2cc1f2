#!/usr/bin/env python3
"""influence != 1 (the filtered[] recurrence, stat.rs:42): ms per pass of the one-wave-per-ctg kernel on the
ctgs of one A. thaliana-sized chromosome, next to the influence == 1 pass over the same bytes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

eng = engine.Engine(0)
ctgs = synth.gen_ctgs("1", synth.chromosome(30_427_671, 1), piece=500000)
ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
for infl in (1.0, 0.5, 0.0):
    plan = engine.WavePlan(eng, ss, 100, 10, 100, 3.0, infl, flags=_lib.WAVE_PEAKS)
    plan.run()
    eng.sync()
    t0 = time.perf_counter()
    reps = 20 if infl == 1.0 else 2
    for _ in range(reps):
        plan.run()
    eng.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"influence {infl}: {len(ctgs)} ctgs, {plan.total_windows} windows, {ms:.3f} ms per pass, "
          f"{plan.total_windows / ms / 1e3:.1f} M windows/s, {plan.peaks().size} peaks")
    plan.close()
