"""Parity at BASELINE.json's full sizes.

configs[1] (S288c, 12 Mb) and configs[2] (A. thaliana, 120 Mb) are small enough for the CPU
oracle to process completely (seconds), so the GPU peaks are compared record by record.
configs[3] geometry (step 1) is checked on one 60-Mb chromosome the same way, plus the
size-independent properties: identical results for every tiling / kernel variant, and for a
ctg processed alone versus inside a batch."""
import numpy as np
import pytest

from gams_amd import _lib, engine, synth
from oracle import oracle as ora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = engine.Engine(0)
    yield e
    e.close()


def oracle_peaks(ctgs, size, step, lag, thr):
    out = []
    for c, ctg in enumerate(ctgs):
        cnt, _, sig = ora.wave_windows(ctg["seq"], size, step, lag, thr, 1.0)
        idx = np.flatnonzero(sig)
        rec = np.zeros(idx.size, _lib.PEAK_DTYPE)
        rec["ctg"], rec["window"], rec["gc_count"], rec["signal"] = c, idx, cnt[idx], sig[idx]
        out.append(rec)
    return np.concatenate(out) if out else np.zeros(0, _lib.PEAK_DTYPE)


def gpu_peaks(eng, ss, size, step, lag, thr, tile=0):
    plan = engine.WavePlan(eng, ss, size, step, lag, thr, 1.0, flags=_lib.WAVE_PEAKS, tile_windows=tile)
    plan.run()
    pk = plan.peaks()
    n = plan.total_windows
    plan.close()
    return pk, n


def test_config1_s288c_full_genome(eng):
    ctgs = synth.genome_ctgs(synth.S288C_LENGTHS, 500000)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    exp = oracle_peaks(ctgs, 100, 10, 100, 3.0)
    for tile in (0, 1024, 2048, 3072, 5120, 1536):  # W = 4 / 8 / 12 / 20; 1536: the general (prefix-array) kernel
        pk, n = gpu_peaks(eng, ss, 100, 10, 100, 3.0, tile)
        assert n == 1215506
        assert np.array_equal(pk, exp), tile
    # four passes in flight (the bench's mode; the default tile becomes W = 8 here): every held pass
    plan = engine.WavePlan(eng, ss, 100, 10, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    plan.set_depth(4)
    plan.run_n(103)
    for age in range(4):
        plan.select(age)
        assert np.array_equal(plan.peaks(), exp), age
    plan.close()
    ss.close()


def test_config2_atha_full_genome(eng):
    ctgs = synth.genome_ctgs(synth.ATHA_LENGTHS, 500000)
    assert sum(len(c["seq"]) for c in ctgs) > 119_000_000
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    pk, n = gpu_peaks(eng, ss, 100, 10, 100, 3.0)
    exp = oracle_peaks(ctgs, 100, 10, 100, 3.0)
    assert np.array_equal(pk, exp)
    assert 0.002 < pk.size / n < 0.05              # peak density like results/Atha.md:217-221
    # a ctg alone gives the same records as inside the batch
    k = len(ctgs) // 2
    s1 = engine.SeqSet(eng, [ctgs[k]["seq"]])
    one, _ = gpu_peaks(eng, s1, 100, 10, 100, 3.0)
    ref = pk[pk["ctg"] == k].copy()
    ref["ctg"] = 0
    assert np.array_equal(one, ref)
    s1.close()
    ss.close()


def test_config3_step1_geometry_one_chromosome(eng):
    chrom = synth.chromosome(60_000_000, 77)
    ctgs = synth.gen_ctgs("77", chrom, piece=1000000)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    pk, n = gpu_peaks(eng, ss, 100, 1, 100, 3.0)
    assert n > 59_000_000
    exp = oracle_peaks(ctgs, 100, 1, 100, 3.0)
    assert np.array_equal(pk, exp)
    for tile in (1024, 5120, 7168, 4096):          # W = 4, 20, 28 (the default here), the general kernel
        other, _ = gpu_peaks(eng, ss, 100, 1, 100, 3.0, tile)
        assert np.array_equal(other, pk), tile
    ss.close()


def test_guard_band_margin_on_the_atha_genome(eng):
    """on the configs[2] genome the default guard band (safety 1.5), the bare bound
    (safety 1.0) and the all-exact run give identical peaks (= the oracle's, checked above), and the
    default run sends fewer than 1e-3 of the windows down the exact path (stat.rs:36-38)."""
    ctgs = synth.genome_ctgs(synth.ATHA_LENGTHS, 500000)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    plan = engine.WavePlan(eng, ss, 100, 10, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    plan.run()
    ref = plan.peaks()
    n = plan.total_windows
    n_default = plan.exact_count()
    assert 0 < n_default < 1e-3 * n
    plan.set_guard(1.0, False)
    plan.run()
    assert np.array_equal(plan.peaks(), ref)
    assert plan.exact_count() <= n_default
    plan.close()
    # all-exact at full size is minutes of one-window-per-wave work: the largest chromosome's first 40 ctgs
    sub = ctgs[:40]
    s2 = engine.SeqSet(eng, [c["seq"] for c in sub])
    p2 = engine.WavePlan(eng, s2, 100, 10, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    p2.set_guard(1.5, True)
    p2.run()
    got = p2.peaks()
    assert np.array_equal(got, ref[ref["ctg"] < 40])
    assert 0.99 * (p2.total_windows - 100 * len(sub)) <= p2.exact_count() <= p2.total_windows - 100 * len(sub)
    p2.close()
    s2.close()
    ss.close()


def test_config2_sw_on_the_atha_genome(eng):
    """BASELINE configs[2], the `fsw` leg (src/cmd_gams/sw.rs:141-184): 1e5 point features (SURVEY 8d C3,
    seed+1) over the A. thaliana-shaped genome through the multi-handle host path.  Full text equality with
    the oracle on the first chromosome's share (> 2.5e4 features, ~1e6 rows); on all features the
    size-independent properties: per feature one M row then L1..Ln then R1..Rm, consecutive distances,
    100-bp windows abutting, everything inside the ctg, gc fields rounded to 4 places in [0, 1]."""
    from gams_amd import host

    ctgs = synth.genome_ctgs(synth.ATHA_LENGTHS, 500000)
    pos = synth.point_features(ctgs, 100000)
    feats = [[(f"feature:{c['id']}:{i + 1}", int(p), int(p)) for i, p in enumerate(ps)] for c, ps in zip(ctgs, pos)]
    n_feat = sum(len(f) for f in feats)
    assert 99_000 < n_feat < 101_000
    engines = [eng, engine.Engine(0)]
    try:
        text = host.sw_multi(engines, ctgs, feats)
    finally:
        engines[1].close()
    rows = text.splitlines()
    assert len(rows) > 40 * n_feat * 0.97                     # <= 41 rows per feature, fewer only at ctg edges
    # ---- oracle text for chromosome 1 (the first ctgs) ----
    first = [i for i, c in enumerate(ctgs) if c["chr_id"] == ctgs[0]["chr_id"]]
    assert sum(len(feats[i]) for i in first) > 25_000
    exp = "".join(ora.sw_proc_ctg(ctgs[i]["chr_id"], ctgs[i]["chr_start"], ctgs[i]["chr_end"], ctgs[i]["seq"], feats[i])
                  for i in first if feats[i])
    n_exp = exp.count("\n")
    assert "\n".join(rows[:n_exp]) + "\n" == exp
    # ---- properties on every row ----
    by_ctg = {c["id"]: c for c in ctgs}
    cur, seen_feat, expect_sn = None, 0, 0
    prev_type, prev_dist, prev_range = None, 0, None
    for r in rows:
        f = r.split("\t")
        assert len(f) == 9
        sid, rg, typ, dist = f[0], f[1], f[2], int(f[3])
        fid, sn = sid[3:].rsplit(":", 1)
        chr_id, se = rg.split(":")
        s, e = (int(x) for x in se.split("-"))
        ctg = by_ctg[fid.split(":", 1)[1].rsplit(":", 1)[0]]
        assert ctg["chr_start"] <= s <= e <= ctg["chr_end"] and chr_id == ctg["chr_id"]
        if fid != cur:
            cur, seen_feat, expect_sn = fid, seen_feat + 1, 1
            assert typ == "M" and dist == 0
            assert e - s + 1 in (99, 100) or ctg["chr_start"] == s or ctg["chr_end"] == e   # clipped at a ctg edge
            m_range = (s, e)
        else:
            assert typ in ("L", "R") and e - s + 1 == 100
            if typ == prev_type:
                assert dist == prev_dist + 1
                assert (e + 1 == prev_range[0]) if typ == "L" else (s == prev_range[1] + 1)
            else:
                assert dist == 1 and (prev_type, typ) in (("M", "L"), ("M", "R"), ("L", "R"))
                assert (e + 1 == m_range[0]) if typ == "L" else (s == m_range[1] + 1)
        assert int(sn) == expect_sn
        expect_sn += 1
        prev_type, prev_dist, prev_range = typ, dist, (s, e)
        assert f[8] == ""                                     # rg_count: declared, never filled (sw.rs:167)
        for k, v in enumerate(f[4:8]):
            x = float(v)
            assert x != x or (0.0 <= x <= 1.0) or (k == 3 and x >= 0.0)   # cv = stddev / mean may exceed 1
            assert len(v.split(".")[-1]) <= 4 or "e" in v               # round(.., 4), Rust {} formatting
    assert seen_feat == n_feat


def test_config3_grch38_step1_sharded_over_eight_handles(eng):
    """BASELINE configs[3] at full size: the GRCh38-shaped genome (24 chromosomes, 3.09 Gb, piece 1000000 ->
    2,937 ctgs), size 100 / step 1 -> 3.09e9 windows.  The ctgs are LPT-sharded by window count over 8
    handles the way `bench.py --workload GRCh38-step1 --gpus 8` shards them over 8 GPUs (device
    k % device_count: one GPU on the test box); every shard runs on its own handle, and the shards' peaks
    must equal (a) the oracle on every 30th ctg, record by record, and (b) the same ctgs' peaks when the
    whole genome is one batch on one handle (sharding invariance).  GAMS_C3_SCALE shrinks the chromosomes."""
    import os
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from gams_amd import shard

    scale = float(os.environ.get("GAMS_C3_SCALE", "1.0"))
    lengths = [int(x * scale) for x in synth.GRCH38_LENGTHS]
    ctgs = synth.genome_ctgs(lengths, 1000000)
    if scale == 1.0:
        assert len(ctgs) == 2937
    weights = [len(c["seq"]) - 100 + 1 for c in ctgs]
    owner = shard.lpt_assign(weights, 8)
    loads = [sum(w for w, o in zip(weights, owner) if o == r) for r in range(8)]
    assert max(loads) <= 1.02 * sum(loads) / 8
    # (b) one batch, one handle
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    whole, n = gpu_peaks(eng, ss, 100, 1, 100, 3.0)
    ss.close()
    assert n == sum(weights)
    assert 0.01 < whole.size / n < 0.05
    # eight shards, eight handles; a shard's ctgs keep their genome order, so its records are the
    # subsequence of the whole run that belongs to its ctgs
    n_dev = max(1, torch.cuda.device_count())
    for r in range(8):
        mine = np.asarray([i for i, o in enumerate(owner) if o == r], np.uint32)
        e = engine.Engine(r % n_dev)
        s = engine.SeqSet(e, [ctgs[i]["seq"] for i in mine])
        pk, _ = gpu_peaks(e, s, 100, 1, 100, 3.0)
        s.close()
        e.close()
        pk = pk.copy()
        pk["ctg"] = mine[pk["ctg"]]
        assert np.array_equal(pk, whole[np.isin(whole["ctg"], mine)]), r
        del pk
    # (a) the oracle on every 30th ctg, in parallel on the host
    sample = list(range(0, len(ctgs), 30))
    with ThreadPoolExecutor(16) as ex:
        exp = list(ex.map(lambda i: oracle_peaks([ctgs[i]], 100, 1, 100, 3.0), sample))
    first = np.searchsorted(whole["ctg"], np.arange(len(ctgs) + 1))
    for i, rec in zip(sample, exp):
        got = whole[first[i]:first[i + 1]].copy()
        got["ctg"] = 0
        assert np.array_equal(got, rec), i


def test_tapered_launch_equals_plain(eng):
    """gams_wave_plan_set_taper: the same genome with the tile table ending in W = 8 / W = 4 tiles (default for a
    plan of depth 1 over at least a round and a half of workgroups), without them, and with passes in flight:
    identical peak records (and the oracle's, through test_config2 above)."""
    ctgs = synth.genome_ctgs(synth.ATHA_LENGTHS, 500000)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    plan = engine.WavePlan(eng, ss, 100, 10, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    assert plan.kernel_name() == "wave_fast_taper_kernel<100, 10, 100, true>"   # 120 MB > 64 MB: streaming loads
    plan.run()
    tapered = plan.peaks()
    n_exact = plan.exact_count()
    plan.set_taper(0)
    assert plan.kernel_name() == "wave_fast_kernel<12, 100, 10, 100, true, 256>"
    plan.run()
    plain = plan.peaks()
    assert np.array_equal(tapered, plain)
    # (which windows take the exact path depends on the tiling: the guard band uses a per-thread bound on S1)
    assert abs(plan.exact_count() - n_exact) <= 0.2 * n_exact
    plan.set_taper(1)
    plan.set_depth(3)
    plan.run_n(7)
    for age in range(3):
        plan.select(age)
        assert np.array_equal(plan.peaks(), plain), age
    plan.set_depth(1)
    plan.set_taper(-1)
    plan.run()
    assert np.array_equal(plan.peaks(), plain)
    with pytest.raises(_lib.GamsError):
        plan.set_taper(2)
    plan.close()
    ss.close()


def test_step1_rows_and_peaks_over_a_long_tile_table(eng):
    """A 60-Mb chromosome at step 1 in one-wave tiles: 35,000 tiles, so the exclusive prefix over the tile counts (and the
    one inside the device-rows graph) takes the three-launch span form (wave_offsets_sum / base / scan_kernel).  Peaks
    and TSV text must equal those of the same pass in 256-thread tiles (8,500 tiles: the one-workgroup prefix), whose
    peaks test_config3_step1_geometry_one_chromosome holds against the oracle."""
    ctgs = synth.gen_ctgs("1", synth.chromosome(60_000_000, 1), piece=1000000)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    got = {}
    for threads in (0, 256):
        plan = engine.WavePlan(eng, ss, 100, 1, 100, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
        plan.set_threads(threads)
        assert plan.kernel_name().endswith(", 64>") == (threads == 0)
        plan.rows_setup([c["chr_id"] for c in ctgs], [c["chr_start"] for c in ctgs], 0.2)
        for rep in range(2):                     # (the second pass copies the text speculatively)
            plan.run()
            plan.rows_begin()
            text, off = plan.rows_end()
        got[threads] = (bytes(text), off.copy(), plan.peaks().copy())
        plan.close()
    ss.close()
    assert got[0][2].size > 500_000
    assert np.array_equal(got[0][2], got[256][2])
    assert got[0][0] == got[256][0] and np.array_equal(got[0][1], got[256][1])


def test_step5_two_wave_tiles_on_a_chromosome(eng):
    """size 100 / step 5 / lag 200 (the reference's own benchmark parameters, doc/benchmark/Atha.md:55) over a 60-Mb
    chromosome: a peaks-only plan runs W = 20 tiles of two waves there; every peak against the oracle, and against the
    same pass in four-wave tiles."""
    ctgs = synth.gen_ctgs("5", synth.chromosome(60_000_000, 5), piece=1000000)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    plan = engine.WavePlan(eng, ss, 100, 5, 200, 3.0, 1.0, flags=_lib.WAVE_PEAKS)
    assert plan.kernel_name() == "wave_fast_kernel<20, 100, 5, 0, false, 128>"
    plan.run()
    pk = plan.peaks().copy()
    plan.set_threads(256)
    assert plan.kernel_name() == "wave_fast_kernel<20, 100, 5, 0, false, 256>"
    plan.run()
    assert np.array_equal(plan.peaks(), pk)
    plan.close()
    ss.close()
    assert np.array_equal(pk, oracle_peaks(ctgs, 100, 5, 200, 3.0))
