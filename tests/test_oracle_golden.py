"""Pins the CPU oracle against every golden vector / known answer the reference
holds for the hot path.  CPU only.

Citations are file:line under the reference (wang-q/gams @ 2024-10-22).
"""
import numpy as np
import pytest

import helpers
from oracle import oracle as ora


# --- libs/stat.rs:58-81 thresholding_sample (lag 30, threshold 5, influence 0) ---
def test_thresholding_sample():
    data = [
        1.0, 1.0, 1.1, 1.0, 0.9, 1.0, 1.0, 1.1, 1.0, 0.9,
        1.0, 1.1, 1.0, 1.0, 0.9, 1.0, 1.0, 1.1, 1.0, 1.0,
        1.0, 1.0, 1.1, 0.9, 1.0, 1.1, 1.0, 1.0, 0.9, 1.0,
        1.1, 1.0, 1.0, 1.1, 1.0, 0.8, 0.9, 1.0, 1.2, 0.9,
        1.0, 1.0, 1.1, 1.2, 1.0, 1.5, 1.0, 3.0, 2.0, 5.0,
        3.0, 2.0, 1.0, 1.0, 1.0, 0.9, 1.0, 1.0, 3.0, 2.6,
        4.0, 3.0, 3.2, 2.0, 1.0, 1.0, 0.8, 4.0, 4.0, 2.0,
        2.5, 1.0, 1.0, 1.0,
    ]
    exp = [
        0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
        0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
        0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
        0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
        0, 0, 0, 0, 0, 1, 0, 1, 1, 1,
        1, 1, 0, 0, 0, 0, 0, 0, 1, 1,
        1, 1, 1, 1, 0, 0, 0, 1, 1, 1,
        1, 0, 0, 0,
    ]
    assert ora.thresholding_algo(data, 30, 5.0, 0.0).tolist() == exp


def test_thresholding_panics_when_short():
    with pytest.raises(ValueError):
        ora.thresholding_algo([0.1] * 5, 10, 3.0, 1.0)


# --- libs/window.rs:58-76 test_center_sw ---
@pytest.mark.parametrize("parent,start,end,exp", [
    ((1, 9999), 500, 500, ((451, 549), "M", 0, 3)),
    ((1, 9999), 500, 800, ((600, 699), "M", 0, 3)),
    ((1, 9999), 101, 101, ((52, 150), "M", 0, 2)),
    ((10001, 19999), 10101, 10101, ((10052, 10150), "M", 0, 2)),
])
def test_center_sw(parent, start, end, exp):
    w = ora.center_sw(parent[0], parent[1], start, end, 100, 1)
    assert (w[0][0], w[0][1]) == exp[0]
    assert w[0][2] == exp[1]
    assert w[0][3] == exp[2]
    assert len(w) == exp[3]


# --- libs/window.rs:126-147 test_center_resize ---
@pytest.mark.parametrize("parent,span,resize,exp", [
    ((1, 500), (201, 201), 100, (152, 250)),
    ((1, 500), (200, 200), 100, (151, 249)),
    ((1, 500), (200, 201), 100, (151, 250)),
    ((1, 500), (199, 201), 100, (150, 249)),
    ((1, 500), (199, 202), 100, (151, 250)),
    ((1, 500), (100, 301), 100, (151, 250)),
    ((1, 500), (1, 1), 100, (1, 50)),
    ((1, 500), (500, 500), 100, (451, 500)),
    ((1001, 1500), (1200, 1201), 100, (1151, 1250)),
])
def test_center_resize(parent, span, resize, exp):
    assert ora.center_resize(parent[0], parent[1], span[0], span[1], resize) == exp


# --- tests/cli.rs:483-495 test_gc_stat; libs/utils.rs:131-134 round doctest ---
def test_gc_stat():
    m, s, c = ora.gc_stat([0.5, 0.5])
    assert (m, s, c) == (0.5, 0.0, 0.0)
    m, s, c = ora.gc_stat([0.4, 0.5, 0.5, 0.6])
    assert m == pytest.approx(0.5, rel=1e-6)
    assert s == pytest.approx(0.0816, rel=1e-6)
    assert c == pytest.approx(0.1633, rel=1e-6)


def test_round():
    assert ora.round_(4.364, 2) == np.float32(4.36)
    assert ora.round_(4.368, 2) == np.float32(4.37)


# --- tests/cli.rs:160-171 (commented out upstream, still informative) ---
def test_gc_content_small(s288c):
    sub = s288c["I"][999:1010]
    assert sub == b"ATACAATTATA"
    assert ora.gc_content(sub) == np.float32(1.0) / np.float32(11.0)
    assert ora.gc_content(s288c["I"][999:1002]) == 0.0


def test_fmt_f32():
    for v, s in [(0.18, "0.18"), (0.3, "0.3"), (0.0, "0"), (1.0, "1"), (0.0816, "0.0816"),
                 (1e-7, "0.0000001"), (12.5, "12.5"), (float("nan"), "NaN"), (float("inf"), "inf"),
                 (100.0, "100"), (0.1 + 0.2, "0.3")]:
        assert ora.fmt_f32(v) == s
    # every k/100 prints the way Rust prints it (shortest round trip == the 2-digit decimal)
    for k in range(101):
        txt = ora.fmt_f32(float(np.float32(k) / np.float32(100)))
        assert float(txt) == pytest.approx(k / 100, abs=1e-9) and len(txt) <= 4


# --- tests/cli.rs:112-146 command_gen: piece 100000 -> ctg:I:1, ctg:I:2, ctg:Mito:1 ---
def test_gen_ctgs(s288c):
    ctgs = []
    for chr_id in ("I", "Mito"):
        ctgs += helpers.gen_ctgs(chr_id, s288c[chr_id], piece=100000)
    assert [c["id"] for c in ctgs] == ["ctg:I:1", "ctg:I:2", "ctg:Mito:1"]
    assert [c["range"] for c in ctgs] == ["I:1-100000", "I:100001-230218", "Mito:1-85779"]
    assert len(s288c["I"]) == 230218 and len(s288c["Mito"]) == 85779


# --- tests/S288c/I.peaks.tsv: README.md:155-163, piece 500000 (BASELINE config 1) ---
def test_wave_golden_peaks(s288c):
    ctgs = helpers.gen_ctgs("I", s288c["I"], piece=500000)
    assert len(ctgs) == 1
    c = ctgs[0]
    out = "#range\tgc_content\tsignal\n" + ora.wave_proc_ctg(
        c["chr_id"], c["chr_start"], c["chr_end"], c["seq"],
        size=100, step=10, lag=100, threshold=3.0, influence=1.0, coverage=0.2)
    golden = "\n".join(helpers.read_lines("I.peaks.tsv")) + "\n"
    assert out == golden
    rows = out.splitlines()[1:]
    assert len(rows) == 116
    assert sum(r.endswith("\t1") for r in rows) == 43      # README.md:166-170
    assert sum(r.endswith("\t-1") for r in rows) == 73
    cnt, gc, sig = ora.wave_windows(c["seq"], 100, 10, 100, 3.0, 1.0)
    assert cnt.size == 23012
    assert int((sig == 1).sum()) == 104 and int((sig == -1).sum()) == 278


# --- tests/cli.rs:332-364 command_wave at piece 100000 ---
def test_wave_piece_100000(s288c):
    ctgs = helpers.gen_ctgs("I", s288c["I"], piece=100000)
    out = "#range\tgc_content\tsignal\n"
    for c in ctgs:
        out += ora.wave_proc_ctg(c["chr_id"], c["chr_start"], c["chr_end"], c["seq"])
    assert len(out.splitlines()) == 116
    assert "I:7551-7650\t" in out
    assert "I(+):11551-11740\t" in out


def _locate(idx, chr_id, start, end):
    """utils.rs:7-22 find_one_idx: Lapper::find(start, end).next()."""
    if chr_id not in idx:
        return ""
    ivs = idx[chr_id]
    starts = np.array([i[0] for i in ivs], np.uint32)
    stops = np.array([i[1] for i in ivs], np.uint32)
    k = ora.lapper_find_first(starts, stops, start, end)
    return ivs[k][2] if k >= 0 else ""


def _all_ctgs(s288c):
    ctgs = []
    for chr_id in ("I", "Mito"):
        ctgs += helpers.gen_ctgs(chr_id, s288c[chr_id], piece=100000)
    return ctgs


def _read_range(lines, idx):
    """utils.rs:39-67 read_range incl. the and_modify/or_default quirk: the first
    range seen for each ctg only creates the (empty) bucket."""
    buckets = {}
    for ln in lines:
        rg = helpers.parse_range(ln.split("\t")[0])
        if rg is None:
            continue
        ctg_id = _locate(idx, *rg)
        if not ctg_id:
            continue
        if ctg_id in buckets:
            buckets[ctg_id].append(rg)
        else:
            buckets[ctg_id] = []
    return dict(sorted(buckets.items()))


# --- tests/cli.rs:384-424 command_locate ---
def test_locate(s288c):
    idx = helpers.ctg_index(_all_ctgs(s288c))
    hits = []
    for q in ["I:1000-1100", "II:1000-1100", "Mito:1000-1100"]:
        ctg = _locate(idx, *helpers.parse_range(q))
        if ctg:
            hits.append((q, ctg))
    assert len(hits) == 2 and hits[0][1] == "ctg:I:1" and all(h[0] != "II:1000-1100" for h in hits)
    lines = helpers.read_lines("spo11_hot.rg")
    assert len(lines) == 79
    located = [_locate(idx, *helpers.parse_range(ln)) for ln in lines]
    located = [x for x in located if x]
    assert len(located) == 71
    assert "ctg:I:1" in located and "ctg:Mito:1" not in located


# --- tests/cli.rs:235-253, 285-304: "There are 69 rgs/features in this file" ---
def test_read_range_drop_first_quirk(s288c):
    idx = helpers.ctg_index(_all_ctgs(s288c))
    buckets = _read_range(helpers.read_lines("spo11_hot.rg"), idx)
    assert sum(len(v) for v in buckets.values()) == 69


# --- tests/cli.rs:426-454 command_locate_count ---
def test_locate_count(s288c):
    idx = helpers.ctg_index(_all_ctgs(s288c))
    buckets = _read_range(helpers.read_lines("SK1.snp.rg"), idx)
    out = []
    for q in ["I:1000-2000", "II:1001-2000", "Mito:1000-2000"]:
        chr_id, s, e = helpers.parse_range(q)
        ctg = _locate(idx, chr_id, s, e)
        if not ctg:
            continue
        rgs = buckets.get(ctg, [])
        starts = np.sort(np.array([r[1] for r in rgs], np.uint32))          # redis.rs:291-294
        stops = np.sort(np.array([r[2] + 1 for r in rgs], np.uint32))
        out.append(f"{q}\t{ora.lapper_count(starts, stops, s, e)}")
    assert out == ["I:1000-2000\t12", "Mito:1000-2000\t0"]


# --- tests/cli.rs:456-481 command_anno ---
def test_anno(s288c):
    sets = helpers.read_runlists("intergenic.json")
    ctgs = {c["id"]: c for c in _all_ctgs(s288c)}
    lines = helpers.read_lines("ctg.range.tsv")
    out = [lines[0] + "\tintergenicProp"]
    for ln in lines[1:]:
        parts = ln.split("\t")
        ctg = ctgs[parts[0]]
        chr_id, s, e = helpers.parse_range(parts[1])
        prop = 0.0
        if chr_id in sets:
            lo, hi = sets[chr_id]
            prop = ora.anno_prop(lo, hi, ctg["chr_start"], ctg["chr_end"], s, e)
        out.append(f"{ln}\t{prop:.4f}")
    assert len(out) == 4 and len(out[0].split("\t")) == 8
    txt = "\n".join(out)
    assert "85779\t0.0000" in txt
    assert "130218\t0.1072" in txt


# --- tests/cli.rs:306-330 command_sw (structure only: numeric values are not pinned upstream) ---
def test_sw_structure(s288c):
    all_ctgs = _all_ctgs(s288c)
    idx = helpers.ctg_index(all_ctgs)
    buckets = _read_range(helpers.read_lines("spo11_hot.rg"), idx)
    out = ""
    for c in sorted(all_ctgs, key=lambda c: c["id"]):
        feats = [(f"feature:{c['id']}:{i + 1}", r[1], r[2]) for i, r in enumerate(buckets.get(c["id"], []))]
        out += ora.sw_proc_ctg(c["chr_id"], c["chr_start"], c["chr_end"], c["seq"], feats)
    rows = out.splitlines()
    assert len(rows) > 2000
    assert any(r.startswith("sw:feature:ctg:I:2:32:1\t") for r in rows)
    assert all(len(r.split("\t")) == 9 for r in rows)


# --- semantics the fixtures do not reach (documented as PARITY UNPINNED in oracle/gams_oracle.h) ---
def test_lapper_half_open_semantics():
    starts = np.array([10, 20, 30], np.uint32)
    stops = np.array([11, 21, 31], np.uint32)      # stored points 10, 20, 30 (stop = end+1)
    assert ora.lapper_count(starts, stops, 10, 20) == 1     # query end exclusive: 20 not counted
    assert ora.lapper_count(starts, stops, 10, 21) == 2
    assert ora.lapper_count(starts, stops, 11, 20) == 0
    # a point query sitting on an interval's start is not found (start < qe fails)
    assert ora.lapper_find_first(np.array([100], np.uint32), np.array([201], np.uint32), 100, 100) == -1
    assert ora.lapper_find_first(np.array([100], np.uint32), np.array([201], np.uint32), 101, 101) == 0
