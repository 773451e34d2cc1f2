#!/usr/bin/env python3
"""rocprofv3 --kernel-trace target: ONE pass of influence 0 / threshold 2 over a 30-Mb chromosome, settled by a reader
(the sweeps of guess-and-iterate, in order, with their durations in the trace)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

eng = engine.Engine(0)
ctgs = synth.gen_ctgs("1", synth.chromosome(30_427_671, 1), piece=500000)
ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
plan = engine.WavePlan(eng, ss, 100, 10, 100, float(sys.argv[1]) if len(sys.argv) > 1 else 2.0, 0.0, flags=_lib.WAVE_PEAKS)
plan.run()
print(plan.settled(), plan.peaks_count())
