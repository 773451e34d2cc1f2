#!/usr/bin/env python3
"""rocprofv3 target: a few passes of influence 0.5 / 0.0 over one 30-Mb chromosome (see tools/serial_rate.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

eng = engine.Engine(0)
ctgs = synth.gen_ctgs("1", synth.chromosome(30_427_671, 1), piece=500000)
ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
plan = engine.WavePlan(eng, ss, 100, 10, 100, thr, float(sys.argv[1]) if len(sys.argv) > 1 else 0.5, flags=_lib.WAVE_PEAKS)
for _ in range(10):
    plan.run()
eng.sync()
print(plan.peaks().size, plan.kernel_name())
