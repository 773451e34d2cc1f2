#!/usr/bin/env python3
"""Where the end-to-end time of one 384-Mb batch goes (host buffers in -> peaks on the host)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, host, synth  # noqa: E402

eng = engine.Engine(0)
small = len(sys.argv) > 1 and sys.argv[1] == "S288c"
ctgs = (synth.genome_ctgs(synth.S288C_LENGTHS, 500000) if small
        else synth.genome_ctgs(synth.SYNTH384_LENGTHS, 1000000, first_chr_index=500))
seqs = [c["seq"] for c in ctgs]
engine.SeqSet(eng, seqs[:4]).close()
for rep in range(5):
    t = [time.perf_counter()]
    lengths = np.array([len(s) for s in seqs], np.uint32)
    p = C.c_void_p()
    eng.check(eng.lib.gams_seqset_create(eng.h, len(seqs), lengths.ctypes.data, C.byref(p)))
    t.append(time.perf_counter())
    arrs = [np.frombuffer(s, np.uint8) for s in seqs]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    t.append(time.perf_counter())
    eng.check(eng.lib.gams_seqset_upload_all(eng.h, p, ptrs))
    t.append(time.perf_counter())
    eng.sync()
    t.append(time.perf_counter())
    ss = engine.SeqSet.__new__(engine.SeqSet)
    ss.eng, ss.p, ss.lengths = eng, p, lengths
    plan = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS)
    t.append(time.perf_counter())
    plan.run()
    eng.sync()
    t.append(time.perf_counter())
    pk = plan.peaks()
    t.append(time.perf_counter())
    plan.close()
    ss.close()
    t.append(time.perf_counter())
    names = ["seqset_create", "ptr array", "upload_all (queue)", "upload drain", "plan_create", "run+sync", "peaks",
             "destroy"]
    print("  ".join(f"{n} {1e3 * (b - a):.2f}" for n, a, b in zip(names, t, t[1:])), f" total {1e3 * (t[-1] - t[0]):.2f} ms",
          flush=True)
hc = [dict(id=c["id"], chr_id=c["chr_id"], chr_start=c["chr_start"], chr_end=c["chr_end"], seq=c["seq"]) for c in ctgs]
for rep in range(3):
    t0 = time.perf_counter()
    out = host.wave(eng, hc)
    print(f"host.wave total {1e3 * (time.perf_counter() - t0):.2f} ms, {out.count(chr(10))} rows")
