#!/usr/bin/env python3
"""Step 1 on the 384-Mb genome: a pass alone, a pass + its packed peak records in host memory, and (under rocprofv3
--kernel-trace --stats) what the packing kernels cost with a tile per wave (227,000 tiles)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

eng = engine.Engine(0)
ctgs = synth.genome_ctgs(synth.SYNTH384_LENGTHS, 1000000, first_chr_index=500)
ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
for threads in (256, 0):
    plan = engine.WavePlan(eng, ss, 100, 1, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    plan.set_threads(threads)
    for _ in range(3):
        plan.run()
        plan.peaks()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        plan.run()
    eng.sync()
    t1 = time.perf_counter()
    for _ in range(10):
        plan.run()
        n = plan.peaks_count()
    eng.sync()
    t2 = time.perf_counter()
    for _ in range(10):
        plan.run()
        pk = plan.peaks()
    eng.sync()
    t3 = time.perf_counter()
    print(f"{plan.kernel_name()}: pass {(t1 - t0) * 100:.3f} ms, pass + packed count {(t2 - t1) * 100:.3f} ms, "
          f"pass + {pk.size} records in host memory {(t3 - t2) * 100:.3f} ms", flush=True)
    plan.close()
ss.close()
