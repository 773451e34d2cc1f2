#!/usr/bin/env python3
"""influence != 1 (the filtered[] recurrence, stat.rs:42): ms per pass (round 2: one wavefront per ctg; round 3:
speculate-and-repair, wave_repair.hpp) on the ctgs of one A. thaliana-sized chromosome and on the whole
Atha-shaped genome, next to the influence == 1 pass over the same bytes; thresholds from dense to sparse."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

eng = engine.Engine(0)
ctgs = synth.gen_ctgs("1", synth.chromosome(30_427_671, 1), piece=500000)
ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])


def rate(ss, n_ctg, infl, thr=3.0, lag=100, step=10):
    plan = engine.WavePlan(eng, ss, 100, step, lag, thr, infl, flags=_lib.WAVE_PEAKS)
    for _ in range(3):
        plan.run()
    eng.sync()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        plan.run()
        plan.peaks_count()          # influence != 1 settles (more sweeps if needed) when somebody reads the pass
    eng.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    sweeps, serial = plan.settled()
    print(f"influence {infl} threshold {thr} lag {lag} step {step}: {n_ctg} ctgs, {plan.total_windows} windows, {ms:.3f} ms per pass, "
          f"{plan.total_windows / ms / 1e3:.1f} M windows/s, {plan.peaks().size} peaks ({plan.kernel_name()}; "
          f"{sweeps} sweeps{', then the serial recurrence' if serial else ''})", flush=True)
    plan.close()


for infl in (1.0, 0.5, 0.0):
    rate(ss, len(ctgs), infl)
for thr in (2.0, 1.0):
    rate(ss, len(ctgs), 0.5, thr)
    rate(ss, len(ctgs), 0.0, thr)
rate(ss, len(ctgs), 0.5, 3.0, 30)
rate(ss, len(ctgs), 0.5, 3.0, 300)
ss.close()
g = synth.genome_ctgs(synth.ATHA_LENGTHS, 500000)
ss = engine.SeqSet(eng, [c["seq"] for c in g])
for infl in (1.0, 0.5, 0.0):
    rate(ss, len(g), infl)
ss.close()
