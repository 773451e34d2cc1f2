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


def genome_ctgs(lengths, piece, seed=SEED, first_chr_index=1, workers=None):
    """All ctgs of a synthetic genome, chromosome by chromosome (chromosomes are seeded on their
    own, so they are generated on `workers` threads; numpy releases the GIL in the heavy calls)."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    def one(k):
        chrom = chromosome(lengths[k], first_chr_index + k, seed)
        return gen_ctgs(str(first_chr_index + k), chrom, piece=piece)

    if workers is None:
        workers = min(16, os.cpu_count() or 1)
    workers = max(1, min(workers, len(lengths)))
    if workers == 1:
        parts = [one(k) for k in range(len(lengths))]
    else:
        # longest chromosomes first, results back in chromosome order
        order = sorted(range(len(lengths)), key=lambda k: -lengths[k])
        with ThreadPoolExecutor(workers) as ex:
            done = dict(zip(order, ex.map(one, order)))
        parts = [done[k] for k in range(len(lengths))]
    return [c for part in parts for c in part]


def layout_ctgs(lengths, piece, min_len=5000, first_chr_index=1):
    """The ctgs `gen` cuts from chromosomes whose only ctg-splitting N runs are the planted 10-kb ones
    (one per 20 Mb), by the --piece rule of gen.rs:108-126 -- no bases generated.
    -> list of (chr_index, serial, chr_start, chr_end), 1-based inclusive, in `gen` order."""
    out = []
    for k, length in enumerate(lengths):
        cuts = []
        for j in range(1, length // 20_000_000 + 1):
            s = j * 20_000_000 - 5000
            if s + 10000 < length:
                cuts.append((s + 1, s + 10000))            # 1-based inclusive N run
        regions, pos = [], 1
        for a, b in cuts:
            regions.append((pos, a - 1))
            pos = b + 1
        regions.append((pos, length))
        serial = 0
        for lo, hi in regions:
            if hi - lo + 1 < min_len:
                continue
            cur, p0 = [], lo
            while hi - p0 + 1 > piece:
                cur.append([p0, p0 + piece - 1])
                p0 += piece
            if not cur:
                cur.append([p0, hi])
            else:
                cur[-1][1] = hi
            for s, e in cur:
                serial += 1
                out.append((first_chr_index + k, serial, s, e))
    return out


def layout_ctg_lengths(lengths, piece, min_len=5000):
    """Lengths of the ctgs `genome_ctgs` cuts, from the planted 10-kb N runs alone (one per 20 Mb)
    and the --piece rule of gen.rs:108-126 -- no bases generated.  Short N runs (1-49 bp) are
    filled by `gen`; two of them abutting into a run of 50 or more (which would split a ctg) is
    rare and ignored here, so this is the layout a sharding test can use at full GRCh38 size."""
    return [e - s + 1 for _, _, s, e in layout_ctgs(lengths, piece, min_len)]


_N_SLOT = 4096      # ctg_bases: at most one short N run per slot, at least 64 bp from the slot's edges


def ctg_bases(chr_index, chr_start, length, seed=SEED):
    """The bases of ONE ctg of a synthetic genome whose layout is `layout_ctgs`: a function of
    (seed, chromosome, start) alone, so a rank generates exactly the ctgs it owns (bench.py strong
    scaling; the ctg is the unit of work, wave.rs:288-299).  Same composition as `chromosome`
    (p(x) on the chromosome coordinate, ~20 % soft-masked runs, short N runs at 1e-5 / bp); the N
    runs sit one per 4096-bp slot, 64 bp clear of the slot's edges, so no two of them merge into a
    run of 50 (which would split the ctg) and none touches the ctg's ends: `gen` over the assembled
    chromosome gives back exactly the layout (tests/test_shard_gloo.py)."""
    rng = np.random.default_rng([seed, chr_index, chr_start])
    out = np.empty(length, np.uint8)
    for b in range(0, length, _CHUNK):
        e = min(b + _CHUNK, length)
        x = np.arange(chr_start - 1 + b, chr_start - 1 + e, dtype=np.float64)
        p = 0.38 + 0.06 * np.sin(2 * np.pi * x / 2300.0) + 0.04 * np.sin(2 * np.pi * x / 97000.0)
        is_gc = rng.random(e - b) < p
        pick = rng.integers(0, 2, e - b, dtype=np.uint8)
        out[b:e] = np.where(is_gc, np.where(pick, 0x47, 0x43), np.where(pick, 0x41, 0x54)).astype(np.uint8)
    n_runs = max(1, int(length * 0.20 / 300))
    starts = rng.integers(0, length, n_runs)
    lens = rng.geometric(1.0 / 300.0, n_runs)
    mark = np.zeros(length + 1, np.int32)
    np.add.at(mark, starts, 1)
    np.add.at(mark, np.minimum(starts + lens, length), -1)
    out[np.cumsum(mark[:-1]) > 0] |= 0x20
    n_slots = length // _N_SLOT
    n_n = min(int(rng.poisson(length * 1e-5)), n_slots)
    if n_n:
        slots = rng.choice(n_slots, n_n, replace=False)
        offs = rng.integers(64, _N_SLOT - 64 - 49, n_n)
        for s, ln in zip((slots * _N_SLOT + offs).tolist(), rng.integers(1, 50, n_n).tolist()):
            out[s:s + ln] = 0x4E
    return out


def sharded_genome_ctgs(lengths, piece, rank=0, world=1, size=100, step=10, seed=SEED, first_chr_index=1,
                        workers=None):
    """Strong-scaling genome: the layout is closed form (`layout_ctgs`), ownership is decided on it
    (LPT by window count), and only the owned ctgs' bases are generated.
    -> (this rank's ctg dicts in genome order, rank loads in windows, number of ctgs in the genome)."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    from . import shard

    lay = layout_ctgs(lengths, piece, first_chr_index=first_chr_index)
    weights = [max(0, (e - s + 1 - size) // step + 1) for _, _, s, e in lay]
    owner = shard.lpt_assign(weights, world)
    loads = [0] * world
    for w, o in zip(weights, owner):
        loads[o] += w
    mine = [c for c, o in zip(lay, owner) if o == rank]

    def one(c):
        k, serial, s, e = c
        return dict(id=f"ctg:{k}:{serial}", chr_id=str(k), chr_start=s, chr_end=e,
                    seq=ctg_bases(k, s, e - s + 1, seed))

    if workers is None:
        workers = min(16, os.cpu_count() or 1)
    if workers <= 1 or len(mine) < 2:
        return [one(c) for c in mine], loads, len(lay)
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(one, mine)), loads, len(lay)


def point_features(ctgs, n, seed=SEED + 1):
    """n uniform point ranges over the genome (features of `sw`), returned per ctg
    as sorted int32 chromosome coordinates (start == end); positions falling between ctgs are dropped
    the way the reference's loaders drop unlocated ranges."""
    rng = np.random.default_rng(seed)
    total = sum(len(c["seq"]) for c in ctgs)
    out = []
    for c in ctgs:
        k = int(round(n * len(c["seq"]) / total))
        pos = np.sort(rng.integers(c["chr_start"], c["chr_end"] + 1, k)).astype(np.int32)
        out.append(pos)
    return out


def c5_workload(share=8, seed=SEED + 5, n_chr=32, chr_len=1_000_000_000, piece=1_000_000,
                n_rg=100_000_000, n_query=100_000_000, spans_per_chr=1_000_000):
    """SURVEY 8(d) C5 (BASELINE configs[4]) cut to one GPU's share: coordinates only, no sequence.
    32 chromosomes x 1e9 bp in 1-Mb ctgs (32,000 ctgs); 1e8 stored point ranges; 1e8 queries of
    length 1-2000; 1e6 disjoint spans per chromosome, length 100-3000.  `share` = number of GPUs the
    whole is divided over: this rank's slice holds n_chr/share chromosomes and 1/share of
    everything else.  (The survey's span lengths of 100-3000 cannot be disjoint at 1e6 spans per
    1e9 bp -- they would cover 155 % -- so the spans here are 100-900 long, one in every 1000-bp
    slot: half of every chromosome is covered.)  Returns a dict of numpy arrays:
      ctg index (groups = ctgs): rg_off u64[n_ctg+1], rg_start/rg_stop u32 (stop = end+1, redis.rs:291-294)
      chr index (groups = chrs): ctg_off u64[n_chr+1], ctg_start/ctg_stop u32 (redis.rs:245-248)
      queries: q_chr u32, q_ctg u32 (the ctg holding q_start), q_start/q_end u32 (inclusive end)
      spans: sp_off u64[n_chr+1], sp_lo/sp_hi i32 (inclusive, sorted, disjoint)"""
    rng = np.random.default_rng(seed)
    chrs = max(1, n_chr // share)
    per_chr = chr_len // piece
    n_ctg = chrs * per_chr
    m = n_rg // share
    nq = n_query // share
    # stored point ranges: uniform over the slice, grouped by ctg
    g = np.sort(rng.integers(0, n_ctg, m, dtype=np.int64))
    rg_off = np.searchsorted(g, np.arange(n_ctg + 1)).astype(np.uint64)
    ctg_first = (g % per_chr) * piece + 1                                  # chr coordinate of the ctg's first base
    rg_start = (ctg_first + rng.integers(0, piece, m)).astype(np.uint32)
    # ctgs per chromosome
    k = np.arange(n_ctg, dtype=np.int64)
    ctg_start = ((k % per_chr) * piece + 1).astype(np.uint32)
    ctg_stop = (ctg_start.astype(np.int64) + piece).astype(np.uint32)      # chr_end + 1
    ctg_off = (np.arange(chrs + 1, dtype=np.uint64) * np.uint64(per_chr))
    # queries
    q_chr = rng.integers(0, chrs, nq, dtype=np.int64)
    q_start = rng.integers(1, chr_len - 2000, nq, dtype=np.int64)
    q_end = q_start + rng.integers(0, 2000, nq, dtype=np.int64)           # length 1..2000
    q_ctg = q_chr * per_chr + (q_start - 1) // piece
    # runlist set: disjoint spans per chromosome
    sp_lo, sp_hi = [], []
    slot = chr_len // spans_per_chr                                        # one span inside every slot
    base = np.arange(spans_per_chr, dtype=np.int64) * slot + 1
    for _ in range(chrs):
        ln = rng.integers(100, max(101, min(3001, slot - 99)), spans_per_chr, dtype=np.int64)
        ofs = rng.integers(0, slot - ln - 1)
        lo = base + ofs
        sp_lo.append(lo.astype(np.int32))
        sp_hi.append((lo + ln - 1).astype(np.int32))
    sp_off = (np.arange(chrs + 1, dtype=np.uint64) * np.uint64(spans_per_chr))
    return dict(n_chr=chrs, n_ctg=n_ctg, per_chr=per_chr, piece=piece,
                rg_off=rg_off, rg_start=rg_start, rg_stop=(rg_start + np.uint32(1)),
                ctg_off=ctg_off, ctg_start=ctg_start, ctg_stop=ctg_stop,
                q_chr=q_chr.astype(np.uint32), q_ctg=q_ctg.astype(np.uint32),
                q_start=q_start.astype(np.uint32), q_end=q_end.astype(np.uint32),
                sp_off=sp_off, sp_lo=np.concatenate(sp_lo), sp_hi=np.concatenate(sp_hi))
