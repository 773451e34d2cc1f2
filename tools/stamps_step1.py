#!/usr/bin/env python3
"""Phase-by-phase cycle profile of the step-1 wave kernel (W = 28 / 20) on the 384-Mb genome."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stamps import profile  # noqa: E402
from gams_amd import engine, synth  # noqa: E402

eng = engine.Engine(0)
big = synth.genome_ctgs([16_000_000] * 24, 1000000, first_chr_index=500)
profile(eng, big, "384Mb step 1", [0, 5120], step=1, reps=5)
eng.close()
