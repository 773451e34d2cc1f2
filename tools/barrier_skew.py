#!/usr/bin/env python3
"""When do the four waves of a workgroup reach a barrier of the wave kernel (SKEW_STEP=1, the default, or 10)?  Experimental builds
(tools/ab/bar{1,2,3}.so: lane 0 of wave w stores the shader clock into stamp slot 12 + w in front of the barrier that
ends phase 1 / phase 2 / at the end of phase 3).  Prints, over all workgroups of one 384-Mb launch: the spread between the
first and the last wave to arrive, and how long after the PREVIOUS stamp of thread 0 each wave arrived."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

STEP = int(os.environ.get("SKEW_STEP", "1"))
ctgs = synth.genome_ctgs([16_000_000] * 24, 1000000, first_chr_index=500)
for path, prev, nxt in zip(sys.argv[1:4], (0, 2, 3), (1, 3, 4)):
    eng = engine.Engine(0, lib=_lib.bind(os.path.abspath(path), strict=False))
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    plan = engine.WavePlan(eng, ss, 100, STEP, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    plan.set_threads(256)
    plan.set_taper(0)
    for _ in range(3):
        plan.run()
    eng.sync()
    eng.check(eng.lib.gams_wave_plan_set_stamps(eng.h, plan.p, 1))
    plan.run()
    eng.sync()
    buf = np.zeros(120000 * 16, np.uint64)
    eng.check(eng.lib.gams_wave_stamps_raw(eng.h, plan.p, buf.ctypes.data, buf.size))
    st = buf.reshape(-1, 16)
    st = st[(st[:, 10] != 0) & (st[:, 12] != 0)].astype(np.int64)
    arr = st[:, 12:16]
    spread = arr.max(axis=1) - arr.min(axis=1)
    since = arr - st[:, prev:prev + 1]
    wait0 = st[:, nxt] - arr[:, 0]
    print(f"{os.path.basename(path)}: {len(st)} workgroups; phase (stamp {prev} -> {nxt}) median {np.median(st[:, nxt] - st[:, prev]):.0f} cycles")
    print("   first-to-last arrival, cycles: percentiles 10/50/90/99", np.percentile(spread, [10, 50, 90, 99]).round(0))
    print("   arrival after the previous stamp, per wave (median):", np.median(since, axis=0).round(0),
          " last wave (median):", np.median(since.max(axis=1)).round(0), " first:", np.median(since.min(axis=1)).round(0))
    print("   wave 0: arrival -> released (median)", np.median(wait0).round(0))
    plan.close()
    ss.close()
    eng.close()
