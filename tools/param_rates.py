#!/usr/bin/env python3
"""Pass time over the 384-Mb genome for parameter sets beside the baked ones (which kernel, windows/s, bytes/s)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

eng = engine.Engine(0)
big = synth.genome_ctgs(synth.SYNTH384_LENGTHS, 1000000, first_chr_index=500)
ss = engine.SeqSet(eng, [c["seq"] for c in big])
for size, step, lag in [(100, 10, 100), (100, 10, 200), (100, 5, 200), (100, 20, 50), (100, 1, 100), (100, 1, 200), (100, 2, 100),
                        (50, 10, 100), (50, 5, 100), (200, 10, 100), (200, 20, 100), (500, 50, 100), (1000, 100, 50), (100, 100, 20),
                        (64, 8, 64), (150, 15, 100), (100, 10, 1000), (30, 3, 30)]:
    try:
        plan = engine.WavePlan(eng, ss, size, step, lag, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    except _lib.GamsError as e:
        print(f"{size}/{step}/{lag}: {e}")
        continue
    plan.run_n(10)
    eng.sync()
    t = []
    for _ in range(5):
        eng.timer_start()
        plan.run_n(5)
        t.append(eng.timer_stop() / 5)
    ms = float(np.median(t))
    nw = plan.total_windows
    print(f"size {size:4d} step {step:3d} lag {lag:4d}: {ms * 1e3:8.1f} us, {nw / ms / 1e6:8.1f} G windows/s, {384e6 / ms / 1e6:6.0f} GB/s of bases  "
          f"{plan.kernel_name()}", flush=True)
    plan.close()
