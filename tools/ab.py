#!/usr/bin/env python3
"""Interleaved A/B timing of builds of libgams_gpu in ONE process on ONE device
(cdna_hip_programming.md rule 24).  usage: tools/ab.py a.so b.so [c.so ...] [--tiles 1024,3072,5120]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--tiles", default="0")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--step", type=int, default=10)
args = ap.parse_args()
tiles = [int(t) for t in args.tiles.split(",")]

small = synth.genome_ctgs(synth.S288C_LENGTHS, 500000)
big = synth.genome_ctgs(synth.SYNTH384_LENGTHS, 1000000, first_chr_index=500)
atha = [synth.genome_ctgs(synth.ATHA_LENGTHS, 500000, first_chr_index=1 + 1000 * g) for g in range(3)]
arms = []
for path in args.libs:
    lib = _lib.bind(os.path.abspath(path), strict=False)
    eng = engine.Engine(0, lib=lib)
    for name, ctgs in (("S288c", small), ("384Mb", big)):
        ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
        for tw in tiles:
            plan = engine.WavePlan(eng, ss, 100, args.step, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS, tile_windows=tw)
            arms.append(dict(lib=os.path.basename(path), wl=name, tw=tw, eng=eng, plan=plan, ss=ss, t=[]))
    # three Atha-shaped genomes, launches rotate over them (every launch streams from HBM)
    sets = [engine.SeqSet(eng, [c["seq"] for c in g]) for g in atha]
    for tw in tiles:
        plans = [engine.WavePlan(eng, ss, 100, args.step, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS, tile_windows=tw)
                 for ss in sets]

        class Rot:
            def __init__(self, plans):
                self.plans, self.i, self.total_windows = plans, 0, plans[0].total_windows

            def run(self):
                self.plans[self.i % 3].run()
                self.i += 1

        arms.append(dict(lib=os.path.basename(path), wl="Atha3", tw=tw, eng=eng, plan=Rot(plans), ss=sets, t=[]))
for a in arms:
    for _ in range(3):
        a["plan"].run()
    a["eng"].sync()
for r in range(args.rounds):
    for a in arms:
        a["eng"].timer_start()
        for _ in range(args.reps):
            a["plan"].run()
        a["t"].append(a["eng"].timer_stop() / args.reps * 1e3)
for a in arms:
    t = np.array(a["t"])
    nw = a["plan"].total_windows
    print(f"{a['wl']:6s} tile={a['tw']:5d} {a['lib']:28s} median {np.median(t):8.2f} us  min {t.min():8.2f} us  "
          f"{nw / np.median(t) / 1e3:7.1f} Gwin/s  {nw * args.step / np.median(t) / 1e3:6.0f} GB/s")
