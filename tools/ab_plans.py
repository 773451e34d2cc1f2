#!/usr/bin/env python3
"""Interleaved A/B of plan-level settings on ONE seqset in ONE process: the arms differ only in the environment
knobs in force when their plan was created (e.g. GAMS_TILE_ORDER=0 vs 1), and read the very same bytes -- two
seqsets of the same content land on different physical pages and differ by up to 5 % on their own.
usage: tools/ab_plans.py KNOB=a KNOB=b [...] [--workload 384|Atha|S288c] [--step 10] [--rounds 9] [--reps 30]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("arms", nargs="+", help="KNOB=value[,KNOB2=value2] per arm")
ap.add_argument("--workload", default="384")
ap.add_argument("--step", type=int, default=10)
ap.add_argument("--rounds", type=int, default=9)
ap.add_argument("--reps", type=int, default=30)
args = ap.parse_args()

eng = engine.Engine(0)
if args.workload == "384":
    ctgs = synth.genome_ctgs(synth.SYNTH384_LENGTHS, 1000000, first_chr_index=500)
elif args.workload == "Atha":
    ctgs = synth.genome_ctgs(synth.ATHA_LENGTHS, 500000)
else:
    ctgs = synth.genome_ctgs(synth.S288C_LENGTHS, 500000)
ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
plans = []
for arm in args.arms:
    kv = dict(x.split("=", 1) for x in arm.split(","))
    for k, v in kv.items():
        os.environ[k] = v
    plans.append(engine.WavePlan(eng, ss, 100, args.step, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS))
    for k in kv:
        del os.environ[k]
ref = None
for p in plans:
    for _ in range(5):
        p.run()
    pk = p.peaks().copy()
    assert ref is None or np.array_equal(pk, ref)
    ref = pk
t = [[] for _ in plans]
for r in range(args.rounds):
    for i, p in enumerate(plans):
        eng.timer_start()
        for _ in range(args.reps):
            p.run()
        t[i].append(eng.timer_stop() / args.reps * 1e3)
for arm, p, ts in zip(args.arms, plans, t):
    ts = np.array(ts)
    print(f"{args.workload:6s} {arm:32s} {p.kernel_name():44s} median {np.median(ts):8.2f} us  min {ts.min():8.2f} us  "
          f"{p.total_windows * args.step / np.median(ts) / 1e3:6.0f} GB/s")
