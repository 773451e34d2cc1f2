#!/usr/bin/env python3
"""Is the host's launch rate the limit at 4 passes in flight?  T host threads each drive their own
handles (streams); ctypes releases the GIL inside the calls."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctgs = synth.genome_ctgs(synth.S288C_LENGTHS, 500000)
seqs = [c["seq"] for c in ctgs]
for n_streams, n_threads in ((4, 1), (4, 2), (4, 4), (8, 2), (8, 4), (8, 8)):
    engs = [engine.Engine(0) for _ in range(n_streams)]
    sets = [engine.SeqSet(e, seqs) for e in engs]
    plans = [engine.WavePlan(e, s, flags=_lib.WAVE_PEAKS, tile_windows=TILE) for e, s in zip(engs, sets)]
    nw = plans[0].total_windows
    for p in plans:
        p.run_n(5)
    for e in engs:
        e.sync()
    K = 2000                      # passes per stream
    per = n_streams // n_threads

    def work(t):
        mine = plans[t * per:(t + 1) * per]
        for _ in range(K // 50):
            for p in mine:
                p.run_n(50)

    best = 1e9
    for rep in range(3):
        ths = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        for e in engs:
            e.sync()
        best = min(best, (time.perf_counter() - t0) / (K * n_streams))
    print(f"tile {TILE}: {n_streams} streams, {n_threads} host threads: {best * 1e6:.2f} us per pass, {nw / best / 1e9:.0f} G windows/s", flush=True)
    for p in plans:
        p.close()
    for s in sets:
        s.close()
    for e in engs:
        e.close()
