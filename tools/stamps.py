#!/usr/bin/env python3
"""Phase-by-phase cycle profile of the wave kernel (diagnostic stamps), per tile size."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import _lib, engine, synth  # noqa: E402

NAMES = ["load+classify", "chunk prefix", "window counts", "z-score", "exact", "outputs", "workgroup"]


def profile(eng, ctgs, label, tiles, step=10, reps=20):
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    for tw in tiles:
        plan = engine.WavePlan(eng, ss, 100, step, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS, tile_windows=tw)
        for _ in range(3):
            plan.run()
        eng.sync()
        eng.timer_start()
        for _ in range(reps):
            plan.run()
        ms = eng.timer_stop() / reps
        m, span = plan.stamps()
        nw = plan.total_windows
        print(f"{label} tile={tw or 'auto'} windows={nw} launch={ms * 1e3:.1f} us  {nw / ms / 1e6:.1f} Gwin/s "
              f"{nw * step / ms / 1e6:.0f} GB/s  {span}")
        print("   " + "  ".join(f"{n}={v:.0f}" for n, v in zip(NAMES, m)))
        plan.close()
    ss.close()


def main():
    eng = engine.Engine(0)
    print(eng.device_info())
    small = synth.genome_ctgs(synth.S288C_LENGTHS, 500000)
    profile(eng, small, "S288c", [0, 1024, 3072, 5120, 2048])
    t = time.time()
    big = synth.genome_ctgs([16_000_000] * 24, 1000000, first_chr_index=500)
    print(f"big genome generated in {time.time() - t:.1f}s")
    profile(eng, big, "384Mb", [0, 1024, 3072, 5120])
    eng.close()


if __name__ == "__main__":
    main()
