"""Deterministic synthetic genomes for bench.py and the size-independent parity tests.

Shapes follow SURVEY.md section 8(d): GC probability
p(x) = 0.38 + 0.06 sin(2 pi x / 2300) + 0.04 sin(2 pi x / 97000), G/C and A/T equiprobable,
~20 % of bases lower-cased in runs (mean 300 bp), N-runs of 1-49 bp at 1e-5 / bp, and one 10 kb
N-run per 20 Mb (which `gen` turns into a ctg break).  The PRNG is numpy's PCG64 seeded with
20241022 ^ chr_index (the survey names xoshiro256**; any fixed generator serves).

`gen_ctgs` restates how the reference cuts chromosomes into ctgs
(src/cmd_gams/gen.rs:81-126: ambiguous-base scan, fill(fill-1), excise(min), --piece chunks, the
last chunk absorbing the remainder).  It only prepares inputs; nothing here is on the hot path.
"""
import numpy as np

SEED = 20241022

# chromosome lengths of the BASELINE.json configs (bp)
S288C_LENGTHS = [230218, 813184, 316620, 1531933, 576874, 270161, 1090940, 562643, 439888, 745751,
                 666816, 1078177, 924431, 784333, 1091291, 948066, 85779]          # sums to 12,157,105
ATHA_LENGTHS = [30427671, 19698289, 23459830, 18585056, 26975502, 366924, 154478]  # 119,667,750
GRCH38_LENGTHS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
                  138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
                  83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]

SYNTH384_LENGTHS = [16_000_000] * 24   # 384 Mb: larger than the 256 MiB Infinity Cache

_CHUNK = 1 << 24


def chromosome(length, chr_index, seed=SEED):
    """uint8 array of `length` bases."""
    rng = np.random.default_rng(seed ^ chr_index)
    out = np.empty(length, np.uint8)
    for b in range(0, length, _CHUNK):
        e = min(b + _CHUNK, length)
        x = np.arange(b, e, dtype=np.float64)
        p = 0.38 + 0.06 * np.sin(2 * np.pi * x / 2300.0) + 0.04 * np.sin(2 * np.pi * x / 97000.0)
        r = rng.random(e - b)
        is_gc = r < p
        pick = rng.integers(0, 2, e - b, dtype=np.uint8)
        # A/T/G/C: G = 0x47, C = 0x43, A = 0x41, T = 0x54
        out[b:e] = np.where(is_gc, np.where(pick, 0x47, 0x43), np.where(pick, 0x41, 0x54)).astype(np.uint8)
    # soft-masked runs: geometric lengths, mean 300, covering ~20 %
    n_runs = max(1, int(length * 0.20 / 300))
    starts = rng.integers(0, length, n_runs)
    lens = rng.geometric(1.0 / 300.0, n_runs)
    mark = np.zeros(length + 1, np.int32)
    np.add.at(mark, starts, 1)
    np.add.at(mark, np.minimum(starts + lens, length), -1)
    lower = np.cumsum(mark[:-1]) > 0
    out[lower] |= 0x20
    # short N runs (kept inside ctgs by --fill 50)
    n_n = rng.poisson(length * 1e-5)
    for s, ln in zip(rng.integers(0, length, n_n), rng.integers(1, 50, n_n)):
        out[s:s + ln] = 0x4E
    # one 10 kb N run per 20 Mb: splits ctgs
    for k in range(1, length // 20_000_000 + 1):
        s = k * 20_000_000 - 5000
        if s + 10000 < length:
            out[s:s + 10000] = 0x4E
    return out


_ACGT = np.zeros(256, bool)
_ACGT[np.frombuffer(b"ACGTacgt", np.uint8)] = True


def gen_ctgs(chr_id, seq, piece=500000, fill=50, min_len=5000):
    """gen.rs:81-157 -> list of dicts(id, chr_id, chr_start, chr_end, seq view)."""
    a = np.frombuffer(seq, np.uint8) if not isinstance(seq, np.ndarray) else seq
    ok = _ACGT[a]                                                       # gen.rs:86-93
    d = np.diff(np.concatenate(([0], ok.view(np.int8), [0])))
    starts = np.flatnonzero(d == 1) + 1
    ends = np.flatnonzero(d == -1)
    filled = []
    for s, e in zip(starts.tolist(), ends.tolist()):                   # fill(fill-1): gen.rs:103
        if filled and s - filled[-1][1] - 1 <= fill - 1:
            filled[-1][1] = e
        else:
            filled.append([s, e])
    filled = [sp for sp in filled if sp[1] - sp[0] + 1 >= min_len]      # excise(min): gen.rs:104
    ctgs, serial = [], 0
    for pos, mx in filled:                                              # gen.rs:108-126
        cur = []
        while mx - pos + 1 > piece:
            cur.append([pos, pos + piece - 1])
            pos += piece
        if not cur:
            cur.append([pos, mx])
        else:
            cur[-1][1] = mx
        for s, e in cur:
            serial += 1
            ctgs.append(dict(id=f"ctg:{chr_id}:{serial}", chr_id=chr_id, chr_start=s, chr_end=e,
                             seq=a[s - 1:e]))
    return ctgs


def genome_ctgs(lengths, piece, seed=SEED, first_chr_index=1):
    """All ctgs of a synthetic genome, chromosome by chromosome."""
    ctgs = []
    for k, length in enumerate(lengths):
        chrom = chromosome(length, first_chr_index + k, seed)
        ctgs += gen_ctgs(str(first_chr_index + k), chrom, piece=piece)
    return ctgs
