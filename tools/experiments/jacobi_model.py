"""CPU model (numpy, f64 prefix sums: convergence behaviour, not bit-exactness) of guess-and-iterate for influence != 1:
signals -> filtered -> signals, from the influence == 1 guess; sweeps until nothing flips.  See gams_amd/csrc/wave_repair.hpp."""
import sys, numpy as np, time
sys.path.insert(0,'/root/repo')
from gams_amd import synth
from oracle import oracle as ora
chrom = synth.chromosome(2_000_000, 1)
seq = chrom[:500_000].tobytes()
size, step, lag = 100, 10, 100
def truth(thr, infl):
    cnt,_,sig = ora.wave_windows(seq, size, step, lag, thr, infl)
    return cnt, sig
def jacobi(cnt, thr, infl, s0, maxit=2000):
    x = (cnt.astype(np.float32)/np.float32(size)).astype(np.float64)
    n = x.size
    sig = s0.copy()
    for it in range(maxit):
        # filtered from sig
        f = x.copy()
        idx = np.flatnonzero(sig)
        for i in idx:   # sequential within runs (in order)
            f[i] = infl*x[i] + (1-infl)*f[i-1]
        # stats for each window i>=lag over f[i-1-lag:i-1] (i==lag: [0,lag))
        c1 = np.concatenate([[0],np.cumsum(f)]); c2 = np.concatenate([[0],np.cumsum(f*f)])
        new = np.zeros(n, np.int32)
        i = np.arange(lag, n)
        a = np.where(i==lag, 0, i-1-lag); b = a+lag
        m = (c1[b]-c1[a])/lag
        v = np.maximum((c2[b]-c2[a]) - lag*m*m, 0)/(lag-1)
        sd = np.sqrt(v)
        hit = np.abs(x[i]-m) > thr*sd
        new[i] = np.where(hit, np.where(x[i]>m,1,-1), 0)
        flips = int((new!=sig).sum())
        sig = new
        if flips==0: return it+1, sig
    return maxit, sig
for thr in (3.0, 2.0, 1.0):
    for infl in (0.5, 0.0):
        cnt, sig_true = truth(thr, infl)
        _, s1 = truth(thr, 1.0)
        t=time.time()
        its, sig = jacobi(cnt, thr, infl, s1.astype(np.int32))
        print(f"thr {thr} infl {infl}: windows {cnt.size} true signals {int((sig_true!=0).sum())} S1 {int((s1!=0).sum())} iterations {its} mismatch_vs_truth {int((sig!=sig_true).sum())} ({time.time()-t:.1f}s)", flush=True)
