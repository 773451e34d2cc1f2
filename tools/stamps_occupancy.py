#!/usr/bin/env python3
"""Phase cycles of the step-1 wave kernel at 1, 2, 4, 8 workgroups per CU (genome size chosen so that one launch is
ONE round of that many workgroups): what a tile costs when it has its SIMDs to itself, and what contention adds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stamps import profile  # noqa: E402
from gams_amd import engine, synth  # noqa: E402

eng = engine.Engine(0)
for per_cu in (1, 2, 4, 8, 32):
    n_tiles = 256 * per_cu
    bases = n_tiles * 7067                       # W = 28 tiles of 7,067 windows at step 1
    ctgs = synth.genome_ctgs([bases // 4] * 4, 1000000, first_chr_index=700)
    profile(eng, ctgs, f"{per_cu} workgroups per CU, step 1", [0], step=1, reps=5)
eng.close()
