#!/usr/bin/env python3
"""Interleaved A/B of plan-level settings on ONE seqset in ONE process: the arms differ only in the environment
knobs in force when their plan was created (e.g. GAMS_TILE_ORDER=0 vs 1), and read the very same bytes -- two
seqsets of the same content land on different physical pages and differ by up to 5 % on their own.
usage: tools/ab_plans.py KNOB=a KNOB=b [...] [--workload 384|Atha|Atha3|S288c] [--step 10] [--rounds 9] [--reps 30]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("arms", nargs="+", help="KNOB=value[,KNOB2=value2] per arm; plan.taper=0 etc. call the plan's setters")
ap.add_argument("--workload", default="384")
ap.add_argument("--step", type=int, default=10)
ap.add_argument("--rounds", type=int, default=9)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--lib", default=None, help="another build of libgams_gpu.so")
ap.add_argument("--flight", action="store_true", help="the plans of an arm on lanes 0, 1, 2 (passes in flight, like bench.py's timed region)")
args = ap.parse_args()

eng = engine.Engine(0, lib=_lib.bind(os.path.abspath(args.lib), strict=False)) if args.lib else engine.Engine(0)
if args.workload == "384":
    genomes = [synth.genome_ctgs(synth.SYNTH384_LENGTHS, 1000000, first_chr_index=500)]
elif args.workload == "Atha":
    genomes = [synth.genome_ctgs(synth.ATHA_LENGTHS, 500000)]
elif args.workload == "Atha3":      # three genomes, launches rotate over them: every launch streams from HBM (bench.py's default)
    genomes = [synth.genome_ctgs(synth.ATHA_LENGTHS, 500000, first_chr_index=1 + 1000 * g) for g in range(3)]
else:
    genomes = [synth.genome_ctgs(synth.S288C_LENGTHS, 500000)]
sets = [engine.SeqSet(eng, [c["seq"] for c in g]) for g in genomes]


class Rot:
    """one arm: a plan per seqset, run() rotates over them"""

    def __init__(self, plans):
        self.plans, self.i, self.total_windows = plans, 0, plans[0].total_windows

    def run(self):
        self.plans[self.i % len(self.plans)].run()
        self.i += 1

    def peaks(self):
        return self.plans[0].peaks()

    def kernel_name(self):
        return self.plans[0].kernel_name()


plans = []
for arm in args.arms:
    kv = dict(x.split("=", 1) for x in arm.split(","))
    env = {k: v for k, v in kv.items() if not k.startswith("plan.")}
    for k, v in env.items():
        os.environ[k] = v
    ps = [engine.WavePlan(eng, ss, 100, args.step, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS) for ss in sets]
    for k, v in kv.items():                     # plan.taper=0|1|-1, plan.tile=3072, plan.depth=2, plan.taper_shape=25050
                                                # (pct4 * 1000 + pct8; was GAMS_TAPER4/8 in the environment): setters of the plan
        if k.startswith("plan."):
            for p in ps:
                getattr(p, "set_" + k[5:])(int(v))
    if args.flight:
        for j, p in enumerate(ps):
            p.set_lane(j % 4)
    plans.append(Rot(ps))
    for k in env:
        del os.environ[k]
ref = None
for p in plans:
    for _ in range(6):
        p.run()
    eng.sync()
    p.i = 0
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
