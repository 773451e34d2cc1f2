#!/usr/bin/env python3
"""Phase-by-phase cycle profile of the step-1 wave kernel for several builds of libgams_gpu (tools/ab.py's arms)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stamps import profile  # noqa: E402
from gams_amd import _lib, engine, synth  # noqa: E402

big = synth.genome_ctgs([16_000_000] * 24, 1000000, first_chr_index=500)
for rep in range(2):
    for path in sys.argv[1:]:
        eng = engine.Engine(0, lib=_lib.bind(os.path.abspath(path), strict=False))
        profile(eng, big, f"{os.path.basename(path)} 384Mb step 1", [0], step=1, reps=5)
        eng.close()
