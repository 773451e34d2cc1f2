"""GPU parity of the host operators (C++ layer over the C ABI) against the oracle's TSV text and
the reference's golden fixtures: wave (merge + rows), sw, locate, locate --count, anno.
These read like the reference's own CLI tests (tests/cli.rs) because the operators mirror it."""
import numpy as np
import pytest

import helpers
from gams_amd import _lib, engine, host
from oracle import oracle as ora

pytestmark = pytest.mark.gpu

HEADER = "#range\tgc_content\tsignal\n"


@pytest.fixture(scope="module")
def eng():
    e = engine.Engine(0)
    yield e
    e.close()


def all_ctgs(s288c, piece=100000):
    ctgs = []
    for chr_id in ("I", "Mito"):
        ctgs += helpers.gen_ctgs(chr_id, s288c[chr_id], piece=piece)
    return ctgs


# ---- wave -------------------------------------------------------------------------------------
def test_wave_golden_I_peaks(eng, s288c):
    """README.md:155-163 -> tests/S288c/I.peaks.tsv, byte for byte."""
    ctgs = helpers.gen_ctgs("I", s288c["I"], piece=500000)
    out = HEADER + host.wave(eng, ctgs, 100, 10, 100, 3.0, 1.0, 0.2)
    assert out == "\n".join(helpers.read_lines("I.peaks.tsv")) + "\n"


def test_command_wave(eng, s288c):
    """tests/cli.rs:332-364."""
    ctgs = helpers.gen_ctgs("I", s288c["I"], piece=100000)
    out = HEADER + host.wave(eng, ctgs)
    assert len(out.splitlines()) == 116
    assert "I:7551-7650\t" in out
    assert "I(+):11551-11740\t" in out


@pytest.mark.parametrize("kw", [
    dict(),
    dict(is_signal=True),
    dict(coverage=1.0),
    dict(coverage=1.5),            # only small overlaps link (size/|overlap| >= 1.5)
    dict(coverage=5.0),
    dict(coverage=50.0),
    dict(size=100, step=1, lag=100),
    dict(size=100, step=1, lag=100, coverage=2.0),
    dict(size=50, step=7, lag=33, threshold=2.5, coverage=0.2),
    dict(size=300, step=10, lag=250),
    dict(influence=0.5),
    dict(influence=0.0, threshold=2.0),
])
def test_wave_rows_equal_oracle(eng, s288c, kw):
    ctgs = all_ctgs(s288c)
    got = host.wave(eng, ctgs, **kw)
    exp = "".join(ora.wave_proc_ctg(c["chr_id"], c["chr_start"], c["chr_end"], c["seq"], **kw) for c in ctgs)
    assert got == exp


def test_wave_from_gz_seq_values(eng, s288c):
    """The path as the reference walks it (wave.rs:134-136 -> redis.rs:142-161): the ctgs' `seq:` values are gzip
    members; workers inflate them straight into a page-locked image of the device buffer and the DMA follows.
    Golden I.peaks.tsv byte for byte from gz values (written by Python's gzip, i.e. not by our encoder), the
    same text as the buffer form on ragged ctgs for 1 / 3 / 16 workers, and the stage clock filled."""
    import gzip

    ctgs = helpers.gen_ctgs("I", s288c["I"], piece=500000)
    for c in ctgs:
        c["gz"] = gzip.compress(bytes(c["seq"]), 1)
    text, st = host.wave_gz(eng, ctgs, 100, 10, 100, 3.0, 1.0, 0.2, threads=4)
    assert HEADER + text.decode() == "\n".join(helpers.read_lines("I.peaks.tsv")) + "\n"
    assert st["inflate_threads"] == 1 and st["peaks"] > 100 and st["total_ms"] > 0     # one ctg: one worker
    ragged = all_ctgs(s288c, piece=30000) + [dict(id="ctg:Z:1", chr_id="Z", chr_start=7, chr_end=7 + 1100 - 1,
                                                  seq=bytes(s288c["Mito"][:1100]))]
    for c in ragged:
        c["gz"] = host.encode_gz(bytes(c["seq"]))
    exp = host.wave(eng, ragged)
    for threads in (1, 3, 16):
        for sync in (False, True):
            text, st = host.wave_gz(eng, ragged, threads=threads, sync=sync)
            assert text.decode() == exp
            assert st["inflate_threads"] == min(threads, len(ragged))
    t2, st2 = host.wave_timed(eng, ragged, sync=True)
    assert t2.decode() == exp and st2["upload_ms"] > 0 and st2["inflate_upload_ms"] == 0
    # a value that inflates to another length than its ctg record says is refused, whichever way it is off
    bad = [dict(c) for c in ragged[:3]]
    bad[1]["gz"] = gzip.compress(bytes(bad[1]["seq"])[:-1], 1)
    with pytest.raises(host.HostError):
        host.wave_gz(eng, bad)
    bad[1]["gz"] = gzip.compress(bytes(bad[1]["seq"]) + b"A", 1)
    with pytest.raises(host.HostError):
        host.wave_gz(eng, bad)
    bad[1]["gz"] = b"\x1f\x8b garbage"
    with pytest.raises(host.HostError):
        host.wave_gz(eng, bad)
    assert host.wave_gz(eng, ragged[:3])[0].decode() == host.wave(eng, ragged[:3])      # the handle is fine afterwards


def test_seqset_image_upload_equals_upload_all(eng, s288c):
    """gams_seqset_layout + gams_seqset_upload_image (a host image of the device buffer, DMA'd in pieces) give
    the same device bytes as gams_seqset_upload_all: same counts and signals; ranges outside the set refused."""
    import ctypes as C

    seqs = [bytes(s288c["I"][:70001]), bytes(s288c["Mito"][:30000]), b"ACGT" * 300, bytes(s288c["I"][100000:100257])]
    ss = engine.SeqSet(eng, seqs)
    lens = np.array([len(x) for x in seqs], np.uint32)
    p2 = C.c_void_p()
    eng.check(eng.lib.gams_seqset_create(eng.h, len(seqs), lens.ctypes.data, C.byref(p2)))
    off = np.zeros(len(seqs), np.uint64)
    nbytes = C.c_uint64()
    eng.check(eng.lib.gams_seqset_layout(eng.h, p2, off.ctypes.data, C.byref(nbytes)))
    assert off[0] == 0 and np.all(off % 256 == 0) and np.all(np.diff(off.astype(np.int64)) >= lens[:-1])
    assert nbytes.value >= int(off[-1]) + int(lens[-1])
    blk = C.c_void_p()
    eng.check(eng.lib.gams_gpu_host_alloc(eng.h, nbytes.value, C.byref(blk)))
    img = np.frombuffer((C.c_uint8 * nbytes.value).from_address(blk.value), np.uint8)
    img[:] = 0x4E
    for o, sq in zip(off, seqs):
        img[int(o):int(o) + len(sq)] = np.frombuffer(sq, np.uint8)
    end = int(off[-1]) + int(lens[-1])
    cuts = [0, 4097, 4097, 65536, end]
    for lo, hi in zip(cuts, cuts[1:]):
        eng.check(eng.lib.gams_seqset_upload_image(eng.h, p2, blk, lo, hi))
    assert eng.lib.gams_seqset_upload_image(eng.h, p2, blk, 10, nbytes.value + 1) == _lib.EINVAL
    assert eng.lib.gams_seqset_upload_image(eng.h, p2, blk, 11, 10) == _lib.EINVAL
    s2 = engine.SeqSet.__new__(engine.SeqSet)
    s2.eng, s2.p, s2.lengths = eng, p2, lens
    for sset in (ss, s2):
        plan = engine.WavePlan(eng, sset, 100, 10, 12, 2.0, 1.0, flags=_lib.WAVE_PEAKS | _lib.WAVE_DENSE)
        plan.run()
        got = [plan.dense(c) for c in range(len(seqs))]
        if sset is ss:
            ref = got
        else:
            for (c0, s0), (c1, s1) in zip(ref, got):
                assert np.array_equal(c0, c1) and np.array_equal(s0, s1)
        plan.close()
    eng.sync()
    eng.lib.gams_gpu_host_free(eng.h, blk)
    s2.close()
    ss.close()


def test_wave_short_ctg_reports_the_reference_panic(eng):
    c = dict(id="ctg:X:1", chr_id="X", chr_start=1, chr_end=400, seq=b"ACGT" * 100)
    with pytest.raises(host.HostError) as ei:
        host.wave(eng, [c])
    assert ei.value.code == 5  # GAMS_ESHORT


# ---- sw ---------------------------------------------------------------------------------------
def bucket_features(s288c, ctgs):
    idx = helpers.ctg_index(ctgs)
    buckets = {}
    for ln in helpers.read_lines("spo11_hot.rg"):
        chr_id, s, e = helpers.parse_range(ln)
        hit = [i for i in idx.get(chr_id, []) if i[0] < e and i[1] > s]
        if not hit:
            continue
        cid = hit[0][2]
        if cid in buckets:
            buckets[cid].append((s, e))
        else:
            buckets[cid] = []          # utils.rs:60-63: the first range only creates the bucket
    return buckets


def test_command_sw(eng, s288c):
    """tests/cli.rs:306-330 (structure) + every numeric field against the oracle."""
    ctgs = all_ctgs(s288c)
    buckets = bucket_features(s288c, ctgs)
    assert sum(len(v) for v in buckets.values()) == 69
    out = ""
    for c in sorted(ctgs, key=lambda c: c["id"]):
        feats = [(f"feature:{c['id']}:{i + 1}", s, e) for i, (s, e) in enumerate(buckets.get(c["id"], []))]
        got = host.sw(eng, c, feats) if feats else ""
        exp = ora.sw_proc_ctg(c["chr_id"], c["chr_start"], c["chr_end"], c["seq"], feats)
        assert got == exp
        out += got
    rows = out.splitlines()
    assert len(rows) > 2000
    assert any(r.startswith("sw:feature:ctg:I:2:32:1\t") for r in rows)


@pytest.mark.parametrize("size,mx,resize", [(100, 20, 500), (100, 1, 100), (50, 5, 333), (10, 40, 5000), (100, 0, 500),
                                            (200, 3, 100)])
def test_sw_random_features(eng, s288c, size, mx, resize):
    c = helpers.gen_ctgs("I", s288c["I"], piece=100000)[1]      # I:100001-230218
    rng = np.random.default_rng(size * 1000 + mx)
    feats = []
    for i in range(300):
        s = int(rng.integers(c["chr_start"], c["chr_end"] + 1))
        e = min(c["chr_end"], s + int(rng.choice([0, 0, 1, 2, 99, 100, 101, 1500])))
        feats.append((f"feature:{c['id']}:{i + 1}", s, e))
    # ctg edges: clipping of M, missing L / R windows
    feats += [(f"feature:{c['id']}:e{k}", p, p) for k, p in enumerate(
        [c["chr_start"], c["chr_start"] + 1, c["chr_start"] + 49, c["chr_start"] + 50, c["chr_start"] + 99,
         c["chr_start"] + 100, c["chr_end"], c["chr_end"] - 1, c["chr_end"] - 49, c["chr_end"] - 50,
         c["chr_end"] - 99, c["chr_end"] - 100])]
    got = host.sw(eng, c, feats, size, mx, resize)
    exp = ora.sw_proc_ctg(c["chr_id"], c["chr_start"], c["chr_end"], c["seq"], feats, size, mx, resize)
    assert got == exp


# ---- locate -----------------------------------------------------------------------------------
def test_command_locate(eng, s288c):
    """tests/cli.rs:384-424."""
    ctgs = all_ctgs(s288c)
    out = host.locate(eng, ctgs, ["I:1000-1100", "II:1000-1100", "Mito:1000-1100"])
    assert out == "I:1000-1100\tctg:I:1\nMito:1000-1100\tctg:Mito:1\n"
    out = host.locate(eng, ctgs, helpers.read_lines("spo11_hot.rg"))
    assert len(out.splitlines()) == 71
    assert "ctg:I:1" in out and "ctg:Mito:1" not in out


def rg_records(eng, ctgs, name):
    """the rg loader (utils.rs:39-67 via cmd_gams/rg.rs:41-77) incl. the drop-first quirk"""
    lines = helpers.read_lines(name)
    ids = host.find(eng, ctgs, [ln.split("\t")[0] for ln in lines])
    seen, recs = set(), []
    for ln, cid in zip(lines, ids):
        if not cid:
            continue
        if cid in seen:
            recs.append((cid, ln.split("\t")[0]))
        seen.add(cid)
    return recs


def test_command_rg_counts(eng, s288c):
    """tests/cli.rs:235-253: 'There are 69 rgs in this file'."""
    assert len(rg_records(eng, all_ctgs(s288c), "spo11_hot.rg")) == 69


def test_command_locate_count(eng, s288c):
    """tests/cli.rs:426-454."""
    ctgs = all_ctgs(s288c)
    recs = rg_records(eng, ctgs, "SK1.snp.rg")
    out = host.locate(eng, ctgs, ["I:1000-2000", "II:1001-2000", "Mito:1000-2000"], count=True, rg_records=recs)
    assert out == "I:1000-2000\t12\nMito:1000-2000\t0\n"


def test_locate_and_count_random_vs_oracle(eng):
    """overlapping stored intervals, point queries on interval edges, queries across ctgs"""
    rng = np.random.default_rng(42)
    ctgs, pos = [], 1
    for k in range(40):                               # one chromosome, 40 ctgs with gaps
        ln = int(rng.integers(5000, 60000))
        ctgs.append(dict(id=f"ctg:1:{k + 1}", chr_id="1", chr_start=pos, chr_end=pos + ln - 1, seq=b""))
        pos += ln + int(rng.integers(0, 3000))
    qs = rng.integers(1, pos + 1000, 5000)
    qe = qs + rng.choice([0, 0, 1, 10, 500, 70000], 5000)
    qs[:40] = [c["chr_start"] for c in ctgs]         # point range on a ctg start: not located (a-16)
    qe[:40] = qs[:40]
    rgs = [f"1:{s}-{e}" if e != s else f"1:{s}" for s, e in zip(qs, qe)]
    starts = np.array([c["chr_start"] for c in ctgs], np.uint32)
    stops = np.array([c["chr_end"] + 1 for c in ctgs], np.uint32)
    got = host.find(eng, ctgs, rgs)
    for r, s, e, g in zip(rgs, qs, qe, got):
        k = ora.lapper_find_first(starts, stops, int(s), int(e))
        assert g == (ctgs[k]["id"] if k >= 0 else ""), r
    assert all(g == "" for g in got[:40])
    # --count against stored intervals of mixed lengths inside each ctg
    recs, per_ctg = [], {}
    for c in ctgs:
        n = int(rng.integers(0, 400))
        a = rng.integers(c["chr_start"], c["chr_end"] + 1, n)
        b = np.minimum(a + rng.choice([0, 0, 0, 5, 300], n), c["chr_end"])
        per_ctg[c["id"]] = (np.sort(a.astype(np.uint32)), np.sort((b + 1).astype(np.uint32)))
        recs += [(c["id"], f"1:{x}-{y}" if y != x else f"1:{x}") for x, y in zip(a, b)]
    out = host.locate(eng, ctgs, rgs, count=True, rg_records=recs).splitlines()
    exp = []
    for r, s, e, g in zip(rgs, qs, qe, got):
        if g:
            st, sp = per_ctg[g]
            exp.append(f"{r}\t{ora.lapper_count(st, sp, int(s), int(e))}")
    assert out == exp


# ---- anno -------------------------------------------------------------------------------------
def test_command_anno(eng, s288c):
    """tests/cli.rs:456-481."""
    ctgs = all_ctgs(s288c)
    import json
    import os

    with open(os.path.join(helpers.S288C, "intergenic.json")) as fh:
        runlists = json.load(fh)
    lines = helpers.read_lines("ctg.range.tsv")
    out = host.anno(eng, ctgs, runlists, lines, header=True, prefix="intergenic", idx_id=1, idx_range=2)
    rows = out.splitlines()
    assert len(rows) == 4 and len(rows[0].split("\t")) == 8
    assert rows[0].endswith("\tintergenicProp")
    assert "85779\t0.0000" in out and "130218\t0.1072" in out


def test_anno_random_vs_oracle(eng):
    rng = np.random.default_rng(9)
    ctgs = [dict(id="ctg:1:1", chr_id="1", chr_start=1, chr_end=400000, seq=b""),
            dict(id="ctg:1:2", chr_id="1", chr_start=400001, chr_end=900000, seq=b""),
            dict(id="ctg:2:1", chr_id="2", chr_start=1001, chr_end=500000, seq=b"")]
    sets = {}
    for chr_id in ("1", "2"):
        cuts = np.sort(rng.choice(np.arange(1, 950000), 4000, replace=False))
        lo, hi = cuts[0::2], cuts[1::2] - 1
        keep = hi >= lo
        sets[chr_id] = (lo[keep].astype(np.int32), hi[keep].astype(np.int32))
    runlists = {k: ",".join(f"{a}-{b}" if a != b else f"{a}" for a, b in zip(*v)) for k, v in sets.items()}
    lines, exp = [], []
    for i in range(3000):
        c = ctgs[int(rng.integers(0, 3))]
        s = int(rng.integers(max(1, c["chr_start"] - 500), c["chr_end"] + 500))
        e = s + int(rng.choice([0, 1, 99, 5000, 200000]))
        chr_id = c["chr_id"] if i % 50 else "3"                       # chr missing from the set -> 0.0000
        rg = f"{chr_id}:{s}-{e}" if e != s else f"{chr_id}:{s}"
        line = f"sw:feature:{c['id']}:{i}:1\t{rg}\tx"
        lines.append(line)
        prop = 0.0
        if chr_id in sets:
            lo, hi = sets[chr_id]
            prop = ora.anno_prop(lo, hi, c["chr_start"], c["chr_end"], s, e)
        exp.append(f"{line}\t{prop:.4f}")
    lines.insert(7, "no contig id here\t1:5-9\tx")                     # dropped (anno.rs:116-119)
    lines.insert(9, f"ctg:1:1\tnot_a_range\tx")                        # dropped (anno.rs:123-125)
    out = host.anno(eng, ctgs, runlists, lines, header=False, idx_id=1, idx_range=2)
    assert out.splitlines() == exp


# ---- gen (first "next" row of SURVEY section 8f) ------------------------------------------------
def test_command_gen(eng, s288c):
    """tests/cli.rs:112-146: piece 100000 -> ctg:I:1, ctg:I:2, ctg:Mito:1; rows of tests/S288c/ctg.tsv."""
    rows = host.gen(eng, "I", s288c["I"], piece=100000) + host.gen(eng, "Mito", s288c["Mito"], piece=100000)
    got = rows.splitlines()
    assert [r.split("\t")[0] for r in got] == ["ctg:I:1", "ctg:I:2", "ctg:Mito:1"]
    assert sorted(got) == sorted(helpers.read_lines("ctg.tsv")[1:])
    one = host.gen(eng, "I", s288c["I"], piece=500000).splitlines()
    assert one == ["ctg:I:1\tI:1-230218\tI\t1\t230218\t+\t230218"]


def test_gen_ambiguous_regions_vs_oracle(eng):
    from gams_amd import synth

    rng = np.random.default_rng(17)
    chrom = synth.chromosome(3_000_000, 41).copy()
    # N runs on both sides of the fill limit, other IUPAC codes, runs at both ends, short valid islands
    for pos, ln in [(0, 30), (5000, 49), (9000, 50), (20000, 51), (100000, 7000), (107100, 40), (107200, 60),
                    (2_999_990, 10), (1_500_000, 1), (1_500_016, 16), (1_600_000 - 3, 35)]:
        chrom[pos:pos + ln] = ord("N")
    chrom[400000:400003] = np.frombuffer(b"RYk", np.uint8)
    chrom[rng.integers(0, chrom.size, 200)] = rng.integers(0, 256, 200)     # arbitrary bytes
    for piece, fill, mn in [(500000, 50, 5000), (100000, 50, 5000), (1000000, 1, 1), (250000, 100, 20000)]:
        rows = host.gen(eng, "7", chrom, piece=piece, fill=fill, min_len=mn).splitlines()
        got = [(int(r.split("\t")[3]), int(r.split("\t")[4])) for r in rows]
        assert got == ora.gen_regions(chrom, piece, fill, mn), (piece, fill, mn)
        assert [r.split("\t")[0] for r in rows] == [f"ctg:7:{i + 1}" for i in range(len(rows))]


def test_read_range_quirk_and_rg_counts(eng, s288c):
    """tests/cli.rs:235-253, 285-304 through the C++ read_range: 79 lines -> 71 located -> 69 kept."""
    ctgs = all_ctgs(s288c)
    recs = host.read_range(eng, ctgs, helpers.read_lines("spo11_hot.rg"))
    assert len(recs) == 69
    assert recs == rg_records(eng, ctgs, "spo11_hot.rg")
    assert {c for c, _ in recs} == {"ctg:I:1", "ctg:I:2"}


def test_locate_seq(eng, s288c):
    """locate --seq (locate.rs:124-134); tests/cli.rs:160-171 names the expected bases."""
    ctgs = all_ctgs(s288c)
    out = host.locate_seq(eng, ctgs, ["I:1000-1002", "I:1000-1010", "II:1000-1100", "I(+):100001-100010"])
    lines = out.splitlines()
    assert lines[:4] == [">I:1000-1002", "ATA", ">I:1000-1010", "ATACAATTATA"]
    assert lines[4] == ">I(+):100001-100010" and lines[5] == s288c["I"][100000:100010].decode()
    assert len(lines) == 6


# ---- peak (second "next" row of SURVEY section 8f; PARITY UNPINNED upstream) ----------------------
def test_command_peak(eng, s288c):
    """`gams peak tests/S288c/I.peaks.tsv` (tests/cli.rs:366-382 only asserts a stderr line).
    Every field is compared with the oracle's restatement of peak.rs:65-158, on the ctg layout the
    fixture was made with (piece 500000, README.md:155-172)."""
    ctgs = helpers.gen_ctgs("I", s288c["I"], piece=500000)
    lines = helpers.read_lines("I.peaks.tsv")
    got = host.peak(eng, ctgs, lines)
    # read_peak (utils.rs:83-116): header row invalid, strand dropped, first peak of each ctg dropped
    peaks = []
    for ln in lines[1:]:
        parts = ln.split("\t")
        chr_id, s, e = helpers.parse_range(parts[0].replace("(+)", ""))
        peaks.append((s, e, parts[2]))
    c = ctgs[0]
    exp = ora.peak_rows(c["id"], c["chr_id"], c["chr_start"], c["chr_end"], c["seq"], peaks[1:])
    assert got == exp
    rows = got.splitlines()
    assert len(rows) == 115 and rows[0].startswith("peak:ctg:I:1:1\tI:3091-3210\t120\t")
    assert all(len(r.split("\t")) == 11 for r in rows)
    # a peak range that leaves its ctg (fixture ranges on the piece-100000 layout) is the reference's panic
    with pytest.raises(host.HostError):
        host.peak(eng, all_ctgs(s288c), lines)


# ---- multi-device host path, exercised with several handles on one GPU --------
def test_wave_multi_handles_equals_single(eng, s288c):
    from gams_amd import synth

    ctgs = all_ctgs(s288c, piece=30000) + synth.genome_ctgs([2_000_000, 700_000], 250000, first_chr_index=90)
    single = host.wave(eng, ctgs)
    e2, e3 = engine.Engine(0), engine.Engine(0)
    try:
        for engines, batch in (([eng, e2], 1 << 30), ([eng, e2, e3], 400_000), ([eng], 100_000)):
            assert host.wave_multi(engines, ctgs, batch_bytes=batch) == single
        assert host.wave_multi([eng, e2], ctgs, is_signal=True, size=50, step=7, lag=33, threshold=2.5) == \
            host.wave(eng, ctgs, is_signal=True, size=50, step=7, lag=33, threshold=2.5)
    finally:
        e2.close()
        e3.close()


def test_command_rg_and_feature_records(eng, s288c):
    """tests/cli.rs:235-253, 285-304: 69 records each; ids, serials and JSON as the loaders SET them."""
    import json

    ctgs = all_ctgs(s288c)
    lines = helpers.read_lines("spo11_hot.rg")
    rg = host.loader_records(eng, ctgs, lines)
    ft = host.loader_records(eng, ctgs, lines, tag="spo11")
    assert len(rg) == 69 and len(ft) == 69
    assert rg[0][0] == "rg:ctg:I:1:1" and ft[0][0] == "feature:ctg:I:1:1"
    assert "feature:ctg:I:2:32" in [k for k, _ in ft]                  # the id tests/cli.rs:325 greps for
    for (k, js), (_, rng) in zip(ft, host.read_range(eng, ctgs, lines)):
        rec = json.loads(js)
        chr_id, s, e = helpers.parse_range(rng)
        assert rec == {"id": k, "range": rng, "length": e - s + 1, "tag": "spo11"}
        assert list(rec) == ["id", "range", "length", "tag"]           # serde field order (data.rs:16-22)
    assert json.loads(rg[5][1]) == {"id": rg[5][0], "range": host.read_range(eng, ctgs, lines)[5][1]}


def test_command_tsv_of_loader_records(eng, s288c):
    """tsv.rs:31-71 over what `gams feature` / `gams rg` stored: header = struct fields (data.rs:16-28)."""
    ctgs = all_ctgs(s288c)
    lines = helpers.read_lines("spo11_hot.rg")
    ft = host.loader_tsv(eng, ctgs, lines, tag="spo11").splitlines()
    rg = host.loader_tsv(eng, ctgs, lines).splitlines()
    assert ft[0] == "id\trange\tlength\ttag" and rg[0] == "id\trange"
    assert len(ft) == 70 and len(rg) == 70
    recs = host.loader_records(eng, ctgs, lines, tag="spo11")
    import json

    for row, (k, js) in zip(ft[1:], recs):
        rec = json.loads(js)
        assert row.split("\t") == [rec["id"], rec["range"], str(rec["length"]), rec["tag"]]


def test_sw_rejects_one_bp_windows(eng, s288c):
    """size or resize 1: center_resize slices [mid+1, mid-1] (window.rs:113-123), an empty span the reference
    then takes min()/max() of; the C ABI reports it instead."""
    c = helpers.gen_ctgs("I", s288c["I"], piece=100000)[0]
    feats = [("feature:x:1", c["chr_start"] + 500, c["chr_start"] + 500)]
    for size, resize in ((1, 500), (100, 1), (0, 500)):
        with pytest.raises(Exception) as ei:
            host.sw(eng, c, feats, size, 20, resize)
        assert "size >= 2" in str(ei.value)


@pytest.mark.parametrize("n_handles", [1, 2, 3])
def test_sw_multi_handles_equals_per_ctg(eng, s288c, n_handles):
    """`gams sw --parallel` over several handles (ctgs split by LPT on their feature counts)"""
    ctgs = all_ctgs(s288c)
    buckets = bucket_features(s288c, ctgs)
    feats = [[(f"feature:{c['id']}:{i + 1}", s, e) for i, (s, e) in enumerate(buckets.get(c["id"], []))] for c in ctgs]
    exp = "".join(ora.sw_proc_ctg(c["chr_id"], c["chr_start"], c["chr_end"], c["seq"], f) for c, f in zip(ctgs, feats) if f)
    engs = [eng] + [engine.Engine(0) for _ in range(n_handles - 1)]
    got = host.sw_multi(engs, ctgs, feats)
    for x in engs[1:]:
        x.close()
    assert got == exp


def test_sw_batch_call_equals_one_call_per_ctg(eng, s288c):
    """gams_gpu_sw_batch (every selected ctg in one launch) against gams_gpu_sw per ctg: same rows in (ctg,
    feature, M/L/R) order, `feature` counted inside each ctg, row_off = where each ctg's rows begin; a
    selection that skips, repeats and reorders ctgs; the size query runs nothing on the device."""
    import ctypes as C
    ctgs = helpers.gen_ctgs("I", s288c["I"], piece=30000)[:7]
    rng = np.random.default_rng(11)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    feats = []
    for c in ctgs:
        n = int(rng.integers(0, 60))
        a = rng.integers(c["chr_start"], c["chr_end"] + 1, n).astype(np.int32)
        b = np.minimum(a + rng.choice([0, 1, 50, 999], n), c["chr_end"]).astype(np.int32)
        feats.append((a, b))
    lib = eng.lib

    def per_ctg(i):
        a, b = feats[i]
        n = C.c_uint64()
        rows = np.zeros(max(a.size * 41, 1), _lib.SW_ROW_DTYPE)
        eng.check(lib.gams_gpu_sw(eng.h, ss.p, i, ctgs[i]["chr_start"], a.ctypes.data, b.ctypes.data, a.size, 100, 20, 500,
                                  rows.ctypes.data, rows.size, C.byref(n)))
        return rows[:n.value]

    for sel in ([0, 1, 2, 3, 4, 5, 6], [5, 2, 2, 6], [3]):
        sel_a = np.array(sel, np.uint32)
        cst = np.array([ctgs[i]["chr_start"] for i in sel], np.int32)
        foff = np.concatenate([[0], np.cumsum([feats[i][0].size for i in sel])]).astype(np.uint64)
        fs = np.ascontiguousarray(np.concatenate([feats[i][0] for i in sel]), np.int32)
        fe = np.ascontiguousarray(np.concatenate([feats[i][1] for i in sel]), np.int32)
        n = C.c_uint64()
        roff = np.zeros(len(sel) + 1, np.uint64)
        args = (eng.h, ss.p, len(sel), sel_a.ctypes.data, cst.ctypes.data, foff.ctypes.data, fs.ctypes.data, fe.ctypes.data,
                100, 20, 500)
        eng.check(lib.gams_gpu_sw_batch(*args, None, 0, roff.ctypes.data, C.byref(n)))          # size query
        exp = [per_ctg(i) for i in sel]
        assert n.value == sum(e.size for e in exp)
        assert np.array_equal(roff, np.concatenate([[0], np.cumsum([e.size for e in exp])]).astype(np.uint64))
        rows = np.zeros(max(n.value, 1), _lib.SW_ROW_DTYPE)
        roff2 = np.zeros_like(roff)
        eng.check(lib.gams_gpu_sw_batch(*args, rows.ctypes.data, rows.size, roff2.ctypes.data, C.byref(n)))
        assert np.array_equal(roff2, roff)
        got = rows[:n.value]
        assert got.tobytes() == (np.concatenate(exp).tobytes() if n.value else b""), sel
    # bad selections
    one = np.array([99], np.uint32)
    z = np.zeros(2, np.uint64)
    n = C.c_uint64()
    assert lib.gams_gpu_sw_batch(eng.h, ss.p, 1, one.ctypes.data, cst.ctypes.data, z.ctypes.data, None, None, 100, 20, 500,
                                 None, 0, None, C.byref(n)) == _lib.EINVAL
    ss.close()


def test_sw_rejects_features_whose_middle_is_outside_the_ctg(eng, s288c):
    """center_resize asks parent.index(mid) (window.rs:110-111): undefined for a non-member; reported instead"""
    c = helpers.gen_ctgs("I", s288c["I"], piece=100000)[1]
    ok = [("f:1", c["chr_start"], c["chr_start"]), ("f:2", c["chr_end"] - 30, c["chr_end"] + 10)]   # middle still inside
    assert host.sw(eng, c, ok).count("\n") > 0
    for s, e in ((c["chr_start"] - 500, c["chr_start"] + 10), (c["chr_end"] - 10, c["chr_end"] + 5000),
                 (c["chr_end"] + 10**6, c["chr_end"] + 10**6 + 5), (500, 400)):
        with pytest.raises(Exception) as ei:
            host.sw(eng, c, [("f:x", s, e)])
        assert "middle outside the ctg" in str(ei.value)


def test_wave_multi_one_handle_per_device(eng, s288c):
    """the multi-device host path with handle k on device k % device_count (hipSetDevice
    per handle); on the one-GPU test box every handle lands on device 0, on an 8-GPU node on all eight."""
    import torch
    from gams_amd import synth

    n_dev = max(1, torch.cuda.device_count())
    ctgs = all_ctgs(s288c, piece=50000) + synth.genome_ctgs([3_000_000], 500000, first_chr_index=70)
    single = host.wave(eng, ctgs)
    engines = [engine.Engine(k % n_dev) for k in range(4)]
    try:
        assert {e.device for e in engines} == set(range(min(4, n_dev)))
        assert host.wave_multi(engines, ctgs) == single
        assert host.wave_multi(engines[:3], ctgs, batch_bytes=500_000) == single
    finally:
        for e in engines:
            e.close()


# ---- wave rows made on the device (gams_wave_rows_*) --------------------------------------------------------
def _host_rows(ctgs, peaks, size, step, coverage):
    """the host layer's merge + formatting over peak records (the path the device rows replace), per ctg"""
    out = []
    for c, ctg in enumerate(ctgs):
        mine = peaks[peaks["ctg"] == c]
        rows = []
        cmin = np.zeros(mine.size, np.int64)
        cmax = np.zeros(mine.size, np.int64)
        merged = np.zeros(mine.size, np.uint8)
        for sgn in (1, -1):
            sel = np.flatnonzero(mine["signal"] == sgn)
            if sel.size:
                a, b, m = host.merge_windows(mine["window"][sel], ctg["chr_start"], size, step, coverage)
                cmin[sel], cmax[sel], merged[sel] = a, b, m
        for i in range(mine.size):
            s = ctg["chr_start"] + int(mine["window"][i]) * step
            e = s + size - 1
            gc = host.fmt_f32(np.float32(mine["gc_count"][i]) / np.float32(size))
            if merged[i]:
                if s != cmin[i]:
                    continue
                rows.append(f"{ctg['chr_id']}(+):{cmin[i]}-{cmax[i]}\t{gc}\t{int(mine['signal'][i])}\n")
            else:
                rng = f"{s}-{e}" if e != s else f"{s}"
                rows.append(f"{ctg['chr_id']}:{rng}\t{gc}\t{int(mine['signal'][i])}\n")
        out.append("".join(rows))
    return out


def test_device_rows_golden_and_against_the_host_merge(eng, s288c):
    """gams_wave_rows_*: the reference's I.peaks.tsv byte for byte from device-made text; the same text as the host's
    merge_ints + formatting over the peak records for steps 1 .. size, ragged ctgs with and without peaks, names of
    different lengths, coverages that link every overlap; coverages that do not are refused (the host merges then)."""
    ctgs = helpers.gen_ctgs("I", s288c["I"], piece=500000)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    plan = engine.WavePlan(eng, ss, 100, 10, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    plan.rows_setup([c["chr_id"] for c in ctgs], [c["chr_start"] for c in ctgs], 0.2)
    plan.run()
    plan.rows_begin()
    text, off = plan.rows_end()
    assert HEADER + text.decode() == "\n".join(helpers.read_lines("I.peaks.tsv")) + "\n"
    assert off.tolist() == [0, len(text)]
    plan.close()
    ss.close()
    quiet = dict(id="ctg:quiet:1", chr_id="quiet", chr_start=5, chr_end=5 + 4000 - 1, seq=b"ACGT" * 1000)    # no peaks at all
    ragged = (all_ctgs(s288c, piece=30000)[:5] + [quiet] + all_ctgs(s288c, piece=30000)[5:]
              + [dict(id="ctg:a-long_name.7:1", chr_id="a-long_name.7", chr_start=1_999_000_001, chr_end=1_999_000_001 + 25000 - 1,
                      seq=bytes(s288c["Mito"][:25000])), quiet])
    ss = engine.SeqSet(eng, [c["seq"] for c in ragged])
    for size, step, lag, thr, cov in [(100, 10, 100, 3.0, 0.2), (100, 1, 100, 3.0, 0.2), (100, 10, 100, 3.0, 1.0),
                                      (100, 100, 20, 2.0, 0.2), (100, 150, 20, 2.0, 0.2), (50, 7, 33, 2.0, 0.5),
                                      (1, 1, 50, 2.0, 0.2), (100, 10, 100, -1.0, 0.2), (100, 3, 40, 1.0, 1.0)]:
        plan = engine.WavePlan(eng, ss, size, step, lag, thr, 1.0, flags=_lib.WAVE_PEAKS)
        plan.rows_setup([c["chr_id"] for c in ragged], [c["chr_start"] for c in ragged], cov)
        for rep in range(2):                      # the second pass copies speculatively, sized by the first
            plan.run()
            plan.rows_begin()
            text, off = plan.rows_end()
        exp = _host_rows(ragged, plan.peaks(), size, step, cov)
        got = [text[int(off[c]):int(off[c + 1])].decode() for c in range(len(ragged))]
        assert got == exp, (size, step, lag, thr, cov)
        assert int(off[-1]) == len(text)
        plan.close()
    plan = engine.WavePlan(eng, ss, 100, 10, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    with pytest.raises(_lib.GamsError) as ei:
        plan.rows_setup([c["chr_id"] for c in ragged], [c["chr_start"] for c in ragged], 1.5)
    assert ei.value.code == _lib.EUNSUPPORTED
    with pytest.raises(_lib.GamsError) as ei:
        plan.rows_begin()
    assert ei.value.code == _lib.ESTATE
    plan.close()
    ss.close()


def test_device_rows_over_tiles_of_one_wave(eng):
    """Step 1 over an S288c-sized synthetic genome: the plan runs a tile per wave (7,000 tiles) and the rows come from
    the same packed records -- device text == the host's merge + formatting, packed peaks == the oracle's on the ctgs
    checked."""
    from gams_amd import synth

    ctgs = synth.genome_ctgs(synth.S288C_LENGTHS, 500000)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    plan = engine.WavePlan(eng, ss, 100, 1, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    assert plan.kernel_name() == "wave_fast_kernel<28, 100, 1, 100, false, 64>"
    plan.rows_setup([c["chr_id"] for c in ctgs], [c["chr_start"] for c in ctgs], 0.2)
    for rep in range(2):
        plan.run()
        plan.rows_begin()
        text, off = plan.rows_end()
    pk = plan.peaks()
    exp = _host_rows(ctgs, pk, 100, 1, 0.2)
    got = [text[int(off[c]):int(off[c + 1])].decode() for c in range(len(ctgs))]
    assert got == exp
    for c in (0, len(ctgs) // 2, len(ctgs) - 1):
        _, _, osig = ora.wave_windows(bytes(ctgs[c]["seq"]), 100, 1, 100, 3.0, 1.0)
        mine = pk[pk["ctg"] == c]
        idx = np.flatnonzero(osig)
        assert np.array_equal(mine["window"], idx) and np.array_equal(mine["signal"], osig[idx])
    plan.close()
    ss.close()


@pytest.mark.parametrize("prm", [dict(), dict(size=100, step=1, lag=100), dict(size=1, step=1, lag=50, threshold=2.0),
                                 dict(size=50, step=7, lag=33, threshold=2.5), dict(size=300, step=10, lag=250),
                                 dict(influence=0.5), dict(influence=0.0, threshold=2.0), dict(size=100, step=150, lag=20)])
def test_signal_text_from_the_device_equals_the_oracle(eng, s288c, prm):
    """gams_wave_signal_text: `wave --signal` rows (one per window, wave.rs:158-168) as text made on the device -- against
    the oracle's text per ctg, with ragged ctgs, names of several lengths, large coordinates, a ctg without any signal,
    single-base windows (the runlist of one position has no dash); then the same through the host operator."""
    quiet = dict(id="ctg:quiet:1", chr_id="quiet", chr_start=5, chr_end=5 + 4000 - 1, seq=b"ACGT" * 1000)
    ctgs = (all_ctgs(s288c, piece=30000)[:4] + [quiet] + all_ctgs(s288c, piece=30000)[4:7]
            + [dict(id="ctg:a-long_name.7:1", chr_id="a-long_name.7", chr_start=1_999_000_001, chr_end=1_999_000_001 + 25000 - 1,
                    seq=bytes(s288c["Mito"][:25000]))])
    kw = dict(size=100, step=10, lag=100, threshold=3.0, influence=1.0)
    kw.update(prm)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    plan = engine.WavePlan(eng, ss, kw["size"], kw["step"], kw["lag"], kw["threshold"], kw["influence"], flags=_lib.WAVE_DENSE)
    for rep in range(2):                       # (the second call reuses the plan's tables and text buffers)
        plan.run()
        text, off = plan.signal_text([c["chr_id"] for c in ctgs], [c["chr_start"] for c in ctgs])
    assert int(off[0]) == 0 and int(off[-1]) == len(text)
    for c, ctg in enumerate(ctgs):
        exp = ora.wave_proc_ctg(ctg["chr_id"], ctg["chr_start"], ctg["chr_end"], ctg["seq"], is_signal=True, **kw)
        assert text[int(off[c]):int(off[c + 1])].decode() == exp, (prm, c)
    plan.close()
    ss.close()
    assert host.wave(eng, ctgs, is_signal=True, **kw) == text.decode()


def test_device_rows_of_several_plans_in_flight(eng, s288c):
    """rows_begin of three plans on three lanes before the first rows_end: every plan gets its own text."""
    batches = [all_ctgs(s288c, piece=40000)[k::3] for k in range(3)]
    sets = [engine.SeqSet(eng, [c["seq"] for c in b]) for b in batches]
    plans = [engine.WavePlan(eng, s, 100, 10, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS) for s in sets]
    for j, (p, b) in enumerate(zip(plans, batches)):
        p.set_lane(j)
        p.set_pipelined(True)
        p.rows_setup([c["chr_id"] for c in b], [c["chr_start"] for c in b], 0.2)
    for rep in range(3):
        for p in plans:
            p.run()
            p.rows_begin()
        for p, b in zip(plans, batches):
            text, off = p.rows_end()
            assert text.decode() == host.wave(eng, b)
    for p in plans:
        p.close()
    for s in sets:
        s.close()


def test_sw_text_from_the_device_equals_the_host_formatter(eng, s288c):
    """gams_gpu_sw_text (rows formatted on the device: the four round4 values printed as m / 10^4 without trailing
    zeros, which is what Rust's `{}` prints for them) against the oracle's text of sw.rs:152-190 and against the host
    formatter over gams_gpu_sw's rows: features at ctg edges (few windows), resize == size (one window per statistic:
    stddev is NaN, cv NaN), long and odd names, ctgs without features, the per-ctg offsets."""
    import ctypes as C

    rng = np.random.default_rng(21)
    ctgs = all_ctgs(s288c, piece=40000)[:6]
    ctgs[2] = dict(ctgs[2], chr_id="chr_with-a.long_name")
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    for size, mx, resize in ((100, 20, 500), (100, 3, 100), (50, 40, 1000), (7, 2, 7)):
        sel, chr_names, cst, foff, fs, fe, ids, feats_per = [], [], [], [0], [], [], [], []
        for i, c in enumerate(ctgs):
            n = 0 if i == 3 else int(rng.integers(1, 40))
            a = np.sort(rng.integers(c["chr_start"], c["chr_end"] + 1, n))
            a[:2] = [c["chr_start"], c["chr_end"]][:n]                      # features on the ctg's first / last base
            b = np.minimum(a + rng.choice([0, 1, 30, 600], n), c["chr_end"])
            feats = [(f"feature:{c['id']}:{j + 1}", int(x), int(y)) for j, (x, y) in enumerate(zip(a, b))]
            feats_per.append(feats)
            sel.append(i)
            chr_names.append(c["chr_id"].encode())
            cst.append(c["chr_start"])
            foff.append(foff[-1] + n)
            fs += [f[1] for f in feats]
            fe += [f[2] for f in feats]
            ids += [f[0].encode() for f in feats]
        sel_a, cst_a, foff_a = np.array(sel, np.uint32), np.array(cst, np.int32), np.array(foff, np.uint64)
        fs_a, fe_a = np.array(fs, np.int32), np.array(fe, np.int32)
        names = (C.c_char_p * len(sel))(*chr_names)
        idp = (C.c_char_p * max(len(ids), 1))(*ids)
        txt, nb, off, nrows = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
        eng.check(eng.lib.gams_gpu_sw_text(eng.h, ss.p, len(sel), sel_a.ctypes.data, names, cst_a.ctypes.data, foff_a.ctypes.data,
                                           fs_a.ctypes.data, fe_a.ctypes.data, idp, size, mx, resize, C.byref(txt), C.byref(nb),
                                           C.byref(off), C.byref(nrows)))
        text = C.string_at(txt.value, nb.value).decode()
        offs = np.frombuffer((C.c_uint64 * (len(sel) + 1)).from_address(off.value), np.uint64)
        assert int(offs[0]) == 0 and int(offs[-1]) == nb.value and text.count("\n") == nrows.value
        for k, c in enumerate(ctgs):
            mine = text[int(offs[k]):int(offs[k + 1])]
            exp = ora.sw_proc_ctg(c["chr_id"], c["chr_start"], c["chr_end"], c["seq"], feats_per[k], size, mx, resize) \
                if feats_per[k] else ""
            assert mine == exp, (size, mx, resize, k)
            if feats_per[k]:
                assert mine == host.sw(eng, c, feats_per[k], size, mx, resize)        # gams_gpu_sw + the host formatter
        if resize == size:
            assert "NaN" in text
    ss.close()
