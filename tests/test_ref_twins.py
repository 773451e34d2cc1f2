"""CPU twins of the C ABI (oracle/gams_ref.h; SURVEY 8b: "every entry has a CPU twin gams_ref_*, same signature"):
each compute entry of include/gams_gpu.h and its twin get the SAME arrays and must return the same outputs.  The
twins are thin adapters over the pinned oracle; the first two tests (CPU) check them against the reference's own
golden answers, so that the GPU tests below compare the ABI with something that is itself anchored."""
import ctypes as C

import numpy as np
import pytest

import helpers
from oracle import oracle as ora

R = ora.ref()


def prm(size=100, step=10, lag=100, thr=3.0, infl=1.0):
    return R.WaveParams(size, step, lag, thr, infl)


def test_twins_reproduce_the_golden_answers(s288c):
    """gams_ref_wave_rows = I.peaks.tsv (README.md:155-163 of the reference); gams_ref_count / gams_ref_cover = the
    known answers of tests/cli.rs (I:1000-2000 counts 12 of the spo11 hot spots; intergenic coverage 0.1072)."""
    seq = np.frombuffer(s288c["I"], np.uint8)
    txt, n = C.c_void_p(), C.c_uint64()
    assert R.gams_ref_wave_rows(b"I", 1, seq.ctypes.data, seq.size, C.byref(prm()), 0.2, C.byref(txt), C.byref(n)) == 0
    rows = C.string_at(txt.value, n.value).decode()
    R.gams_ref_free(txt)
    assert "#range\tgc_content\tsignal\n" + rows == "\n".join(helpers.read_lines("I.peaks.tsv")) + "\n"
    nw = C.c_uint32()
    cnt = np.zeros(23012, np.uint32)
    sig = np.zeros(23012, np.int8)
    assert R.gams_ref_wave(seq.ctypes.data, seq.size, C.byref(prm()), cnt.ctypes.data, sig.ctypes.data, C.byref(nw)) == 0
    assert nw.value == 23012 and int((sig != 0).sum()) > 100
    short = np.frombuffer(b"ACGT" * 100, np.uint8)
    assert R.gams_ref_wave(short.ctypes.data, short.size, C.byref(prm()), None, None, C.byref(nw)) == 5     # GAMS_ESHORT
    # SK1 SNPs on chr I as the idx:rg of one ctg: `locate --count` of I:1000-2000 is 12 (tests/cli.rs:306-330).
    # (read_range drops the first range of every ctg, utils.rs:39-67; chr I is one ctg here and its first SNP is far
    # from 1000-2000 either way.)
    snp = [ln.split(":")[1] for ln in helpers.read_lines("SK1.snp.rg") if ln.startswith("I:")][1:]
    st = np.array([int(x.split("-")[0]) for x in snp], np.uint32)
    sp = np.array([int(x.split("-")[-1]) + 1 for x in snp], np.uint32)           # stop = end + 1 (redis.rs:291-294)
    off = np.array([0, st.size], np.uint64)
    g, qs, qe = np.zeros(1, np.uint32), np.array([1000], np.uint32), np.array([2000], np.uint32)
    out = np.zeros(1, np.int32)
    assert R.gams_ref_count(1, off.ctypes.data, st.ctypes.data, sp.ctypes.data, g.ctypes.data, qs.ctypes.data,
                            qe.ctypes.data, 1, out.ctypes.data) == 0
    assert out[0] == 12


def test_twin_signatures_follow_the_header():
    """every twin's parameter list = its ABI entry's, device objects replaced (checked on the C text)"""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref_h = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "oracle", "gams_ref.h")).read(), flags=re.S)
    twins = dict(re.findall(r"int (gams_ref_[a-z_]+)\(([^;]*?)\);", ref_h, flags=re.S))
    assert set(twins) == {"gams_ref_wave", "gams_ref_wave_peaks", "gams_ref_wave_rows", "gams_ref_wave_signal_text", "gams_ref_sw",
                          "gams_ref_sw_text", "gams_ref_range_gc", "gams_ref_count", "gams_ref_locate", "gams_ref_cover",
                          "gams_ref_valid_spans"}
    abi_h = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "gams_gpu.h")).read(), flags=re.S)
    abi = dict(re.findall(r"int (gams_gpu_[a-z_]+)\(([^;]*?)\);", abi_h, flags=re.S))

    def names(params):
        return [re.split(r"[\s\*]+", p.strip())[-1] for p in params.split(",")]

    # the query columns and outputs keep the ABI's names and order
    for twin, entry, tail in (("gams_ref_count", "gams_gpu_count", 5), ("gams_ref_locate", "gams_gpu_locate", 5),
                              ("gams_ref_cover", "gams_gpu_cover", 7), ("gams_ref_valid_spans", "gams_gpu_valid_spans", 8),
                              ("gams_ref_wave", "gams_gpu_wave", 6)):
        assert names(twins[twin])[-tail:] == names(abi[entry])[-tail:], twin


# ---- the ABI against its twins, on the device ------------------------------------------------------------------
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from gams_amd import engine

    e = engine.Engine(0)
    yield e
    e.close()


@gpu
def test_wave_entries_equal_their_twins(eng, s288c):
    from gams_amd import _lib, engine

    seqs = [np.frombuffer(s288c["I"][:120_000], np.uint8), np.frombuffer(s288c["Mito"][:30_000], np.uint8)]
    for p in (prm(), prm(50, 7, 33, 2.5, 1.0), prm(100, 1, 100, 3.0, 1.0), prm(100, 10, 100, 3.0, 0.5)):
        lp = _lib.WaveParams(p.size, p.step, p.lag, p.threshold, p.influence)
        for sq in seqs:
            n = (sq.size - p.size) // p.step + 1
            a_cnt, a_sig, b_cnt, b_sig = np.zeros(n, np.uint32), np.zeros(n, np.int8), np.zeros(n, np.uint32), np.zeros(n, np.int8)
            na, nb = C.c_uint32(), C.c_uint32()
            eng.check(eng.lib.gams_gpu_wave(eng.h, sq.ctypes.data, sq.size, C.byref(lp), a_cnt.ctypes.data, a_sig.ctypes.data,
                                            C.byref(na)))
            assert R.gams_ref_wave(sq.ctypes.data, sq.size, C.byref(p), b_cnt.ctypes.data, b_sig.ctypes.data, C.byref(nb)) == 0
            assert na.value == nb.value == n and np.array_equal(a_cnt, b_cnt) and np.array_equal(a_sig, b_sig)
        # the batch: compacted peaks
        ss = engine.SeqSet(eng, seqs)
        plan = engine.WavePlan(eng, ss, p.size, p.step, p.lag, p.threshold, p.influence, flags=_lib.WAVE_PEAKS)
        plan.run()
        got = plan.peaks()
        ptrs = (C.c_void_p * 2)(*[s.ctypes.data for s in seqs])
        lens = np.array([s.size for s in seqs], np.uint32)
        npk = C.c_uint64()
        exp = np.zeros(got.size + 10, _lib.PEAK_DTYPE)
        assert R.gams_ref_wave_peaks(2, ptrs, lens.ctypes.data, C.byref(p), exp.ctypes.data, exp.size, C.byref(npk)) == 0
        assert npk.value == got.size and np.array_equal(got, exp[:got.size])
        # ... and the rows as text
        if p.influence == 1.0:
            plan.rows_setup(["I", "Mito"], [1, 5], 0.2)
            plan.rows_begin()
            text, off = plan.rows_end()
            for c, (name, cs) in enumerate((("I", 1), ("Mito", 5))):
                txt, nb_ = C.c_void_p(), C.c_uint64()
                assert R.gams_ref_wave_rows(name.encode(), cs, seqs[c].ctypes.data, seqs[c].size, C.byref(p), 0.2,
                                            C.byref(txt), C.byref(nb_)) == 0
                assert text[int(off[c]):int(off[c + 1])] == C.string_at(txt.value, nb_.value)
                R.gams_ref_free(txt)
        plan.close()
        # ... and the --signal rows of every window (plans with the dense rows)
        plan = engine.WavePlan(eng, ss, p.size, p.step, p.lag, p.threshold, p.influence, flags=_lib.WAVE_DENSE)
        plan.run()
        text, off = plan.signal_text(["I", "Mito"], [1, 5])
        for c, (name, cs) in enumerate((("I", 1), ("Mito", 5))):
            txt, nb_ = C.c_void_p(), C.c_uint64()
            assert R.gams_ref_wave_signal_text(name.encode(), cs, seqs[c].ctypes.data, seqs[c].size, C.byref(p), C.byref(txt),
                                               C.byref(nb_)) == 0
            assert text[int(off[c]):int(off[c + 1])] == C.string_at(txt.value, nb_.value)
            R.gams_ref_free(txt)
        plan.close()
        ss.close()


@gpu
def test_sw_range_gc_and_gen_equal_their_twins(eng, s288c):
    from gams_amd import _lib, engine

    seq = np.frombuffer(s288c["I"][:150_000], np.uint8)
    chr_start = 1001
    rng = np.random.default_rng(4)
    fs = np.sort(rng.integers(chr_start, chr_start + seq.size - 50, 300)).astype(np.int32)
    fe = (fs + rng.integers(0, 40, fs.size)).astype(np.int32)
    ss = engine.SeqSet(eng, [seq])
    cap = fs.size * 41
    a_rows, b_rows = np.zeros(cap, _lib.SW_ROW_DTYPE), np.zeros(cap, _lib.SW_ROW_DTYPE)
    na, nb = C.c_uint64(), C.c_uint64()
    eng.check(eng.lib.gams_gpu_sw(eng.h, ss.p, 0, chr_start, fs.ctypes.data, fe.ctypes.data, fs.size, 100, 20, 500,
                                  a_rows.ctypes.data, cap, C.byref(na)))
    assert R.gams_ref_sw(seq.ctypes.data, seq.size, chr_start, fs.ctypes.data, fe.ctypes.data, fs.size, 100, 20, 500,
                         b_rows.ctypes.data, cap, C.byref(nb)) == 0
    assert na.value == nb.value > 0
    assert a_rows[:na.value].tobytes() == b_rows[:nb.value].tobytes()          # every field, floats bit for bit
    # the same rows as text: gams_gpu_sw_text against its twin
    ids = [f"feature:ctg:I:1:{j + 1}".encode() for j in range(fs.size)]
    id_arr = (C.c_char_p * fs.size)(*ids)
    chr_arr = (C.c_char_p * 1)(b"I")
    sel, cst, foff = np.zeros(1, np.uint32), np.array([chr_start], np.int32), np.array([0, fs.size], np.uint64)
    txt, tb, toff, nr = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
    eng.check(eng.lib.gams_gpu_sw_text(eng.h, ss.p, 1, sel.ctypes.data, chr_arr, cst.ctypes.data, foff.ctypes.data, fs.ctypes.data,
                                       fe.ctypes.data, id_arr, 100, 20, 500, C.byref(txt), C.byref(tb), C.byref(toff), C.byref(nr)))
    rt, rb = C.c_void_p(), C.c_uint64()
    assert R.gams_ref_sw_text(b"I", seq.ctypes.data, seq.size, chr_start, fs.ctypes.data, fe.ctypes.data, id_arr, fs.size, 100, 20,
                              500, C.byref(rt), C.byref(rb)) == 0
    assert nr.value == na.value and C.string_at(txt.value, tb.value) == C.string_at(rt.value, rb.value)
    R.gams_ref_free(rt)
    rs = rng.integers(chr_start, chr_start + seq.size - 2000, 500).astype(np.int32)
    re_ = (rs + rng.integers(0, 1999, rs.size)).astype(np.int32)
    a_gc, b_gc = np.zeros(rs.size, np.float32), np.zeros(rs.size, np.float32)
    eng.check(eng.lib.gams_gpu_range_gc(eng.h, ss.p, 0, chr_start, rs.ctypes.data, re_.ctypes.data, rs.size, a_gc.ctypes.data))
    assert R.gams_ref_range_gc(seq.ctypes.data, seq.size, chr_start, rs.ctypes.data, re_.ctypes.data, rs.size, b_gc.ctypes.data) == 0
    assert a_gc.tobytes() == b_gc.tobytes()
    ss.close()
    chrom = np.frombuffer(s288c["I"], np.uint8).copy()
    chrom[5000:5070] = ord("N")
    chrom[90_000:96_000] = ord("n")
    chrom[200_000:200_030] = ord("R")
    for fill, mn in ((50, 5000), (1, 1), (100, 100_000)):
        lo_a, hi_a, lo_b, hi_b = (np.zeros(64, np.int32) for _ in range(4))
        na, nb = C.c_uint64(), C.c_uint64()
        eng.check(eng.lib.gams_gpu_valid_spans(eng.h, chrom.ctypes.data, chrom.size, fill, mn, lo_a.ctypes.data, hi_a.ctypes.data, 64,
                                               C.byref(na)))
        assert R.gams_ref_valid_spans(chrom.ctypes.data, chrom.size, fill, mn, lo_b.ctypes.data, hi_b.ctypes.data, 64, C.byref(nb)) == 0
        assert na.value == nb.value and np.array_equal(lo_a[:na.value], lo_b[:nb.value]) and np.array_equal(hi_a[:na.value], hi_b[:nb.value])


@gpu
def test_interval_entries_equal_their_twins(eng):
    rng = np.random.default_rng(9)
    sizes = [0, 1, 700, 40, 0, 2500]
    off = np.cumsum([0] + sizes).astype(np.uint64)
    st = np.concatenate([rng.integers(0, 50_000, n) for n in sizes]).astype(np.uint32)
    sp = (st + rng.choice([1, 1, 3, 80, 900], st.size)).astype(np.uint32)
    nq = 20_000
    g = rng.integers(0, len(sizes) + 2, nq).astype(np.uint32)            # two unknown groups too
    qs = rng.integers(0, 51_000, nq).astype(np.uint32)
    qe = (qs + rng.choice([0, 1, 2, 30, 2000], nq)).astype(np.uint32)
    ix = C.c_void_p()
    eng.check(eng.lib.gams_index_create(eng.h, len(sizes), off.ctypes.data, st.ctypes.data, sp.ctypes.data, C.byref(ix)))
    a_cnt, b_cnt = np.zeros(nq, np.int32), np.zeros(nq, np.int32)
    a_hit, b_hit = np.zeros(nq, np.int64), np.zeros(nq, np.int64)
    eng.check(eng.lib.gams_gpu_count(eng.h, ix, g.ctypes.data, qs.ctypes.data, qe.ctypes.data, nq, a_cnt.ctypes.data))
    eng.check(eng.lib.gams_gpu_locate(eng.h, ix, g.ctypes.data, qs.ctypes.data, qe.ctypes.data, nq, a_hit.ctypes.data))
    eng.lib.gams_index_destroy(eng.h, ix)
    assert R.gams_ref_count(len(sizes), off.ctypes.data, st.ctypes.data, sp.ctypes.data, g.ctypes.data, qs.ctypes.data,
                            qe.ctypes.data, nq, b_cnt.ctypes.data) == 0
    assert R.gams_ref_locate(len(sizes), off.ctypes.data, st.ctypes.data, sp.ctypes.data, g.ctypes.data, qs.ctypes.data,
                             qe.ctypes.data, nq, b_hit.ctypes.data) == 0
    assert np.array_equal(a_cnt, b_cnt)
    # equal (start, stop) pairs are interchangeable for Lapper::find().next(): compare the interval found
    assert np.array_equal(a_hit < 0, b_hit < 0)
    k = a_hit >= 0
    assert np.array_equal(st[a_hit[k]], st[b_hit[k]]) and np.array_equal(sp[a_hit[k]], sp[b_hit[k]])
    # spans
    n_sp = [0, 300, 1, 1200]
    soff = np.cumsum([0] + n_sp).astype(np.uint64)
    lo, hi = [], []
    for n in n_sp:
        cuts = np.sort(rng.choice(np.arange(-20_000, 80_000), 2 * n, replace=False))
        lo.append(cuts[0::2])
        hi.append(cuts[1::2] - 1)
    lo = np.concatenate(lo).astype(np.int32)
    hi = np.concatenate(hi).astype(np.int32)
    keep = hi >= lo
    assert keep.all()
    sg = rng.integers(0, len(n_sp) + 1, nq).astype(np.uint32)
    s = rng.integers(-21_000, 81_000, nq).astype(np.int32)
    e = (s + rng.choice([0, 5, 99, 3000], nq)).astype(np.int32)
    cl = (s - rng.choice([0, 10, 10_000], nq)).astype(np.int32)
    ch = (e + rng.choice([-3, 0, 10, 10_000], nq)).astype(np.int32)
    spn = C.c_void_p()
    eng.check(eng.lib.gams_spans_create(eng.h, len(n_sp), soff.ctypes.data, lo.ctypes.data, hi.ctypes.data, C.byref(spn)))
    a_p, b_p = np.zeros(nq, np.float32), np.zeros(nq, np.float32)
    eng.check(eng.lib.gams_gpu_cover(eng.h, spn, sg.ctypes.data, cl.ctypes.data, ch.ctypes.data, s.ctypes.data, e.ctypes.data, nq,
                                     a_p.ctypes.data))
    eng.lib.gams_spans_destroy(eng.h, spn)
    assert R.gams_ref_cover(len(n_sp), soff.ctypes.data, lo.ctypes.data, hi.ctypes.data, sg.ctypes.data, cl.ctypes.data,
                            ch.ctypes.data, s.ctypes.data, e.ctypes.data, nq, b_p.ctypes.data) == 0
    assert a_p.tobytes() == b_p.tobytes()
