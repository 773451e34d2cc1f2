import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gams_amd import _lib, engine, synth
eng = engine.Engine(0)
ctgs = synth.genome_ctgs(synth.S288C_LENGTHS, 500000)
ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
plan = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS)
plan.set_depth(4)
plan.run_n(20); eng.sync(); torch.cuda.synchronize()
for K in (50, 200, 1000):
    for rep in range(4):
        t0 = time.perf_counter(); plan.run_n(K); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
        print(f"K={K}: run_n {1e6*(t1-t0):.0f} us, eng.sync {1e6*(t2-t1):.0f} us, torch sync {1e6*(t3-t2):.0f} us, total/K {1e6*(t3-t0)/K:.2f}")
