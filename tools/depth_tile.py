import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth
eng = engine.Engine(0)
ctgs = synth.genome_ctgs(synth.S288C_LENGTHS, 500000)
ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
for tw in (1024, 2048, 3072):
    plan = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS, tile_windows=tw)
    nw = plan.total_windows
    for depth in (1, 4):
        plan.set_depth(depth)
        plan.run_n(50); eng.sync()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter(); plan.run_n(2000); eng.sync(); best = min(best, (time.perf_counter() - t0) / 2000)
        print(f"tile {tw} depth {depth}: {best*1e6:.2f} us/pass, {nw/best/1e9:.0f} G windows/s", flush=True)
    plan.close()
