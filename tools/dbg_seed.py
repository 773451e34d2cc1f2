import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gams_amd import _lib, engine
from oracle import oracle as ora
from test_gpu_random_params import random_seq
eng = engine.Engine(0)
seed = int(sys.argv[1])
for rep in range(4):
    rng = np.random.default_rng(1000 + seed)
    size = int(rng.choice([1, 7, 10, 50, 64, 100, 100, 100, 128, 200, 255, 256, 300, 1000]))
    step = int(rng.choice([1, 2, 5, 10, 10, 10, 16, 31, 32, 33, 50, 64, 100, 150]))
    lag = int(rng.choice([2, 3, 10, 33, 50, 100, 100, 127, 128, 200, 400]))
    thr = float(rng.choice([0.5, 1.0, 2.0, 2.5, 3.0, 3.0, 3.5, 5.0]))
    infl = float(rng.choice([1.0, 1.0, 1.0, 1.0, 0.5, 0.0]))
    n_ctg = int(rng.integers(1, 5))
    need = size + (lag + 5) * step
    seqs = [random_seq(rng, int(need + rng.integers(0, 40 * need // 10 + 5000))) for _ in range(n_ctg)]
    ss = engine.SeqSet(eng, seqs)
    plan = engine.WavePlan(eng, ss, size, step, lag, thr, infl, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE)
    plan.run()
    pk = plan.peaks()
    exp = []
    for c, s in enumerate(seqs):
        ocnt, _, osig = ora.wave_windows(s, size, step, lag, thr, infl)
        idx = np.flatnonzero(osig)
        exp += [(c, int(i), int(ocnt[i]), int(osig[i])) for i in idx]
    got = [(int(r["ctg"]), int(r["window"]), int(r["gc_count"]), int(r["signal"])) for r in pk]
    bad = [i for i, (a, b) in enumerate(zip(got, exp)) if a != b]
    print(rep, (size, step, lag, thr, infl), "ctgs", [len(s) for s in seqs], "windows", plan.total_windows, "peaks", len(got), len(exp), "bad", bad[:10], [got[i] for i in bad[:4]], [exp[i] for i in bad[:4]])
    pk2 = plan.peaks()
    got2 = [(int(r["ctg"]), int(r["window"]), int(r["gc_count"]), int(r["signal"])) for r in pk2]
    print("   second peaks() equal to expected:", got2 == exp)
    plan.close(); ss.close()
