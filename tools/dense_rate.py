#!/usr/bin/env python3
"""Throughput of the --signal (dense rows: u32 count + i8 signal per window) mode next to the peaks mode."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

eng = engine.Engine(0)
big = synth.genome_ctgs(synth.SYNTH384_LENGTHS, 1000000, first_chr_index=500)
ss = engine.SeqSet(eng, [c["seq"] for c in big])
for step in (10, 1):
    for name, flags in (("peaks", _lib.WAVE_PEAKS), ("dense (--signal)", _lib.WAVE_DENSE), ("both", _lib.WAVE_PEAKS | _lib.WAVE_DENSE)):
        plan = engine.WavePlan(eng, ss, 100, step, 100, 3.0, 1.0, flags=flags)
        if len(sys.argv) > 1:
            plan.set_threads(int(sys.argv[1]))      # step-1 tiles of 64 / 128 / 256 threads
        plan.run_n(100 if step == 10 else 20)
        eng.sync()
        t = []
        for _ in range(5):
            eng.timer_start()
            plan.run_n(10)
            t.append(eng.timer_stop() / 10)
        ms = float(np.median(t))
        nw = plan.total_windows
        out_b = 5 if flags & _lib.WAVE_DENSE else 0
        print(f"step {step:2d} {name:18s}: {ms * 1e3:8.1f} us per pass, {nw / ms / 1e6:7.1f} G windows/s, "
              f"{nw * (step + out_b) / ms / 1e6:7.0f} GB/s algorithmic ({step} B read + {out_b} B written per window)", flush=True)
        plan.close()
