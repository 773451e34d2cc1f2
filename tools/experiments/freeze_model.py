"""influence 0: once `lag` consecutive windows have signalled, filtered[] is one constant for the rest of the ctg (a signalled
window keeps filtered[i-1], an unsignalled one has x == that constant), so every later decision is a function of
(count at the freeze, count of the window).  CPU model (numpy, f64 statistics: behaviour, not bit-exactness) of
guess-and-iterate with that as an extra guess step: sweeps until nothing flips, with and without it."""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from gams_amd import synth
from oracle import oracle as ora

size, step, lag = 100, 10, 100
chrom = synth.chromosome(2_000_000, 1)


def stats_decide(x, f, thr):
    n = x.size
    c1 = np.concatenate([[0], np.cumsum(f)]); c2 = np.concatenate([[0], np.cumsum(f * f)])
    new = np.zeros(n, np.int32)
    i = np.arange(lag, n)
    a = np.where(i == lag, 0, i - 1 - lag); b = a + lag
    m = (c1[b] - c1[a]) / lag
    v = np.maximum((c2[b] - c2[a]) - lag * m * m, 0) / (lag - 1)
    hit = np.abs(x[i] - m) > thr * np.sqrt(v) + 1e-12
    new[i] = np.where(hit, np.where(x[i] > m, 1, -1), 0)
    return new


def fill_forward(x, sig):
    idx = np.where(sig == 0, np.arange(x.size), 0)
    last = np.maximum.accumulate(idx)            # last unsignalled window <= i (window 0 never signals)
    return x[last], last


def jacobi0(x, thr, s0, freeze, maxit=400):
    sig = s0.copy()
    for it in range(maxit):
        f, last = fill_forward(x, sig)
        if freeze:
            run = np.arange(x.size) - last
            hit = np.flatnonzero(run >= lag)
            if hit.size:
                p = hit[0]                       # the run that started at last[p] + 1 has reached `lag` windows
                c = x[last[p]]
                sig[p + 1:] = np.where(x[p + 1:] > c, 1, np.where(x[p + 1:] < c, -1, 0))
                f, last = fill_forward(x, sig)
        new = stats_decide(x, f, thr)
        flips = int((new != sig).sum())
        sig = new
        if flips == 0:
            return it + 1, sig
    return maxit, sig


for off, thr in ((0, 3.0), (0, 2.0), (0, 1.0), (700_000, 2.0), (1_200_000, 2.5)):
    seq = chrom[off:off + 500_000].tobytes()
    cnt, _, truth = ora.wave_windows(seq, size, step, lag, thr, 0.0)
    _, _, s1 = ora.wave_windows(seq, size, step, lag, thr, 1.0)
    x = (cnt.astype(np.float32) / np.float32(size)).astype(np.float64)
    f_true, last = fill_forward(x, truth)
    run = np.arange(x.size) - last
    hit = np.flatnonzero(run >= lag)
    frozen = "none" if hit.size == 0 else f"at window {hit[0]} of {x.size}: filtered takes {np.unique(f_true[hit[0]:]).size} value(s) behind it"
    for freeze in (False, True):
        t = time.time()
        its, sig = jacobi0(x, thr, s1.astype(np.int32), freeze)
        print(f"thr {thr} off {off}: true signals {int((truth != 0).sum())}, freeze {frozen}; "
              f"{'with' if freeze else 'without'} the freeze guess: {its} sweeps, mismatch {int((sig != truth).sum())} ({time.time() - t:.1f}s)", flush=True)
