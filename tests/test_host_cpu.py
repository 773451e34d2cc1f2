"""CPU-only checks of the C++ host layer's bookkeeping (formatting, range grammar)."""
import numpy as np
import pytest

from gams_amd import host
from oracle import oracle as ora


def test_fmt_f32_matches_rust_display_and_the_oracle():
    vals = [0.18, 0.3, 0.0, 1.0, 0.0816, 1e-7, 12.5, 100.0, 0.1633, 0.5, 123456.7, 3.4e38, 1.17e-38]
    vals += [float(np.float32(k) / np.float32(100)) for k in range(101)]
    vals += [float(np.float32(k) / np.float32(257)) for k in range(0, 257, 7)]
    rng = np.random.default_rng(1)
    vals += [float(x) for x in rng.random(500).astype(np.float32)]
    vals += [float(np.round(np.float32(x) * 10000) / np.float32(10000)) for x in rng.random(200)]
    for v in vals:
        assert host.fmt_f32(v) == ora.fmt_f32(v), v
    assert host.fmt_f32(float("nan")) == "NaN"
    assert host.fmt_f32(float("inf")) == "inf"
    assert host.fmt_f32(0.18) == "0.18" and host.fmt_f32(1.0) == "1" and host.fmt_f32(0.0) == "0"


@pytest.mark.parametrize("s,exp", [
    ("I:1000-1100", "I:1000-1100"),
    ("I(+):11551-11740", "I(+):11551-11740"),
    ("Mito:136", "Mito:136"),
    ("S288c.I(-):27070-29557", "S288c.I(-):27070-29557"),
    ("I:5-5", "I:5"),
    ("I", "<invalid>"),
    ("", "<invalid>"),
    ("I:abc", "<invalid>"),
    ("#range", "<invalid>"),
    ("chr-1_x:1_200", "chr-1_x:1-200"),
])
def test_range_grammar(s, exp):
    assert host.range_roundtrip(s) == exp


def test_gzip_framing_of_seq_values():
    """redis.rs:149-161: seq: values are gzip members; any RFC 1952 codec must interoperate."""
    import gzip
    import os

    import helpers

    raw = helpers.load_s288c()["Mito"][:50000]
    assert host.decode_gz(gzip.compress(raw, 1)) == raw          # what flate2 fast() writes
    assert gzip.decompress(host.encode_gz(raw)) == raw           # what flate2 GzDecoder reads
    assert host.decode_gz(host.encode_gz(b"")) == b""
    with open(os.path.join(helpers.S288C, "genome.fa.gz"), "rb") as fh:
        blob = fh.read()
    assert host.decode_gz(blob) == gzip.decompress(blob)         # the reference's own fixture
    with pytest.raises(host.HostError):
        host.decode_gz(b"not a gzip member")


def test_gzip_batches_on_several_threads_and_damaged_members():
    """decode_gz of a batch of seq: values on several host threads (the reference inflates inside its --parallel
    workers, redis.rs:142-161 under wave.rs:288-299); Python's gzip module is the independent codec.  Members
    whose ISIZE trailer lies, truncated members, a second member behind the first (GzDecoder reads one),
    values that do not fit the room given."""
    import gzip
    import struct

    import helpers
    import numpy as np

    seqs = helpers.load_s288c()
    rng = np.random.default_rng(5)
    raws = [bytes(seqs["I"][a:a + n]) for a, n in ((0, 100000), (100000, 130218), (5, 1), (77, 0), (1000, 65536))]
    raws += [bytes(rng.integers(0, 256, 30000, dtype=np.uint8)), b"N" * 200000]
    blobs = [gzip.compress(r, lv) for r, lv in zip(raws, (1, 6, 1, 1, 9, 1, 1))]
    for threads in (1, 3, 16):
        out = host.decode_gz_many(blobs, [len(r) for r in raws], threads)
        assert [o.tobytes() for o in out] == raws
    # a trailer whose ISIZE lies is a corrupt member (flate2, zlib and libdeflate all check it)
    good = gzip.compress(raws[0], 1)
    for lie in (0, 5, 2 * len(raws[0]), 0xFFFFFFFF):
        with pytest.raises(host.HostError):
            host.decode_gz(good[:-4] + struct.pack("<I", lie))
    # a second member behind the first: flate2's GzDecoder reads one member; the single-value form sizes its
    # output from the LAST four bytes (the short second member's ISIZE here) and has to grow
    assert host.decode_gz(good + gzip.compress(b"second member", 1)) == raws[0]
    for bad in (good[:len(good) // 2], good[:17], good[:10] + b"\xff" * 40 + good[50:], b""):
        with pytest.raises(host.HostError):
            host.decode_gz(bad)
    with pytest.raises(host.HostError):
        host.decode_gz_many([good], [len(raws[0]) - 1], 2)                          # does not fit


def test_command_tsv_ctgs_reproduces_the_fixture():
    """tests/cli.rs:214-233 (`gams tsv -s "ctg:*"`: 4 lines, 7 fields, header names) and the rows of
    tests/S288c/ctg.tsv, which the reference made the same way (README.md recipe)."""
    import helpers

    lines = helpers.read_lines("ctg.tsv")
    ctgs = []
    for row in lines[1:]:
        f = row.split("\t")
        ctgs.append(dict(id=f[0], chr_id=f[2], chr_start=int(f[3]), chr_end=int(f[4]), seq=b""))
    out = host.tsv_ctgs(ctgs).splitlines()
    assert len(out) == 4 and len(out[0].split("\t")) == 7
    assert "chr_strand\tlength" in out[0] and any("ctg:I:2" in r for r in out)
    assert out[0] == lines[0]
    assert out[1:] == lines[1:]


def test_command_headers():
    """wave.rs:263-266 and sw.rs:204-216; tests/S288c/I.peaks.tsv starts with the wave header"""
    import helpers

    assert host.header("wave") == "#range\tgc_content\tsignal\n"
    assert host.header("wave").rstrip("\n") == helpers.read_lines("I.peaks.tsv")[0]
    assert host.header("sw").rstrip("\n").split("\t") == ["id", "range", "type", "distance", "gc_content", "gc_mean",
                                                           "gc_stddev", "gc_cv", "rg_count"]


def test_merge_ints_against_the_pairwise_rule():
    """wave.rs:217-252 restated pair by pair (every i<j: non-empty intersection, size/|cap| >= coverage on both
    sides, f32 like the reference; components by union-find) against the host's distance-range merge."""
    rng = np.random.default_rng(12)
    for it in range(300):
        size = int(rng.choice([10, 50, 100, 100, 255, 1000]))
        step = int(rng.choice([1, 5, 10, 10, 33, 100, 150, 999]))
        cov = float(rng.choice([0.05, 0.2, 0.2, 0.5, 1.0, 1.05, 1.5, 3.0]))
        n = int(rng.integers(0, 60))
        w = np.unique(rng.integers(0, 40 + n * rng.choice([1, 3, 30]), n)).astype(np.uint32)
        chr_start = int(rng.integers(1, 10**6))
        cmin, cmax, ing = host.merge_windows(w, chr_start, size, step, cov)
        s = chr_start + w.astype(np.int64) * step
        e = s + size - 1
        par = list(range(w.size))

        def find(x):
            while par[x] != x:
                par[x] = par[par[x]]
                x = par[x]
            return x

        edge = np.zeros(w.size, bool)
        for i in range(w.size):
            for j in range(i + 1, w.size):
                inter = min(e[i], e[j]) - max(s[i], s[j]) + 1
                if inter <= 0:
                    continue
                c = np.float32(size) / np.float32(inter)
                if c >= np.float32(cov):
                    edge[i] = edge[j] = True
                    a, b = find(i), find(j)
                    if a != b:
                        par[b] = a
        roots = [find(i) for i in range(w.size)]
        for i in range(w.size):
            members = [k for k in range(w.size) if roots[k] == roots[i]]
            assert cmin[i] == min(s[k] for k in members) and cmax[i] == max(e[k] for k in members), (it, i)
        assert np.array_equal(ing, edge), it


def test_ctg_json_record_roundtrip():
    """the `ctg:{chr}:{sn}` value `gen` stores (serde_json of data.rs:5-14, redis.rs:127-135): field order and
    number formats of serde_json; the parser takes any field order and whitespace"""
    import helpers
    import json

    for row in helpers.read_lines("ctg.tsv")[1:]:
        f = row.split("\t")
        rec = dict(id=f[0], range=f[1], chr_id=f[2], chr_start=int(f[3]), chr_end=int(f[4]), chr_strand=f[5],
                   length=int(f[6]))
        compact = json.dumps(rec, separators=(",", ":"))           # what serde_json::to_string writes
        assert host.ctg_json_roundtrip(compact) == compact
        shuffled = json.dumps(dict(reversed(list(rec.items()))), indent=2)
        assert host.ctg_json_roundtrip(shuffled) == compact
    with pytest.raises(Exception):
        host.ctg_json_roundtrip('{"range":"I:1-5"}')
