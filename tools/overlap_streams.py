#!/usr/bin/env python3
"""How much do back-to-back independent passes gain from being queued on 2..4 streams?
Each handle owns one compute stream; plans on different handles may overlap on the device."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "S288c"
ctgs = (synth.genome_ctgs(synth.S288C_LENGTHS, 500000) if wl == "S288c"
        else synth.genome_ctgs(synth.SYNTH384_LENGTHS, 1000000, first_chr_index=500))
seqs = [c["seq"] for c in ctgs]
for n_streams in (1, 2, 3, 4):
    engs = [engine.Engine(0) for _ in range(n_streams)]
    sets = [engine.SeqSet(e, seqs) for e in engs]
    plans = [engine.WavePlan(e, s, flags=_lib.WAVE_PEAKS) for e, s in zip(engs, sets)]
    nw = plans[0].total_windows
    for p in plans:
        for _ in range(5):
            p.run()
    for e in engs:
        e.sync()
    best = 1e9
    for rep in range(5):
        K = 400
        t0 = time.perf_counter()
        for k in range(K):
            plans[k % n_streams].run()
        for e in engs:
            e.sync()
        best = min(best, (time.perf_counter() - t0) / K)
    print(f"{wl}: {n_streams} stream(s): {best * 1e6:.2f} us per pass, {nw / best / 1e9:.1f} G windows/s", flush=True)
    for p in plans:
        p.close()
    for s in sets:
        s.close()
    for e in engs:
        e.close()
