"""Wire formats either side of the hot path: bincode 1.3.3 of the `bundle:ctg:` map and of
the rust-lapper `idx:` blobs, RESP2 framing.  PARITY UNPINNED: bincode, rust-lapper and the redis crate
are not in /root/reference and no fixture of the reference holds these bytes; the expected vectors below
are assembled by hand from the crates' published formats (little endian, u64 lengths, fields in declaration
order: src/libs/data.rs:5-14; struct Lapper {intervals, starts, stops, max_len, cov, overlaps_merged})."""
import struct

import pytest

from gams_amd import host


def s(x):          # bincode String
    b = x.encode()
    return struct.pack("<Q", len(b)) + b


def ctg_bytes(c):
    rng = f"{c['chr_id']}:{c['chr_start']}-{c['chr_end']}"
    return (s(c["id"]) + s(c["id"]) + s(rng) + s(c["chr_id"]) + struct.pack("<ii", c["chr_start"], c["chr_end"])
            + s("+") + struct.pack("<i", c["chr_end"] - c["chr_start"] + 1))


CTGS = [dict(id="ctg:I:2", chr_id="I", chr_start=100001, chr_end=230218),
        dict(id="ctg:I:10", chr_id="I", chr_start=900001, chr_end=900500),      # "ctg:I:10" < "ctg:I:2" as strings
        dict(id="ctg:I:1", chr_id="I", chr_start=1, chr_end=100000)]


def test_ctg_bundle_bytes_and_round_trip():
    blob = host.bincode_ctg_bundle(CTGS)
    order = sorted(CTGS, key=lambda c: c["id"].encode())                         # BTreeMap: byte-wise key order
    assert [c["id"] for c in order] == ["ctg:I:1", "ctg:I:10", "ctg:I:2"]
    assert blob == struct.pack("<Q", 3) + b"".join(ctg_bytes(c) for c in order)
    tsv = host.bincode_ctg_bundle_decode(blob)
    assert tsv == host.tsv_ctgs(order)
    assert host.bincode_ctg_bundle([]) == struct.pack("<Q", 0)
    # the S288c fixture: gen's three ctgs (tests/cli.rs:112-146) survive the round trip
    fixture = [dict(id="ctg:I:1", chr_id="I", chr_start=1, chr_end=100000),
               dict(id="ctg:I:2", chr_id="I", chr_start=100001, chr_end=230218),
               dict(id="ctg:Mito:1", chr_id="Mito", chr_start=1, chr_end=85779)]
    assert host.bincode_ctg_bundle_decode(host.bincode_ctg_bundle(fixture)) == host.tsv_ctgs(fixture)


@pytest.mark.parametrize("cut", [0, 7, 8, 20, 57, -1])
def test_ctg_bundle_rejects_damaged_input(cut):
    blob = host.bincode_ctg_bundle(CTGS)
    bad = blob[:cut] if cut >= 0 else blob + b"\0"
    with pytest.raises(host.HostError):
        host.bincode_ctg_bundle_decode(bad)


def test_ctg_bundle_decodes_like_a_btreemap_and_rejects_huge_lengths():
    """bincode feeds the entries to BTreeMap::insert: any order is accepted, a repeated KEY keeps the last
    value, iteration is in key order (what get_bundle_ctg's callers see, src/libs/redis.rs:216-233)."""
    a, b = ctg_bytes(CTGS[0]), ctg_bytes(CTGS[2])
    tsv = host.bincode_ctg_bundle_decode(struct.pack("<Q", 2) + a + b)          # ctg:I:2 before ctg:I:1
    assert tsv == host.tsv_ctgs([CTGS[2], CTGS[0]])
    # same key twice: the later entry wins -- judged on the stored key, not on the value's id
    later = dict(CTGS[0], chr_start=100002)
    dup = s("ctg:I:2") + ctg_bytes(later)[len(s("ctg:I:2")):]
    assert host.bincode_ctg_bundle_decode(struct.pack("<Q", 3) + a + b + dup) == host.tsv_ctgs([CTGS[2], later])
    # a key that differs from its value's id orders the entry
    renamed = s("ctg:A") + ctg_bytes(CTGS[0])[len(s("ctg:I:2")):]
    assert host.bincode_ctg_bundle_decode(struct.pack("<Q", 2) + b + renamed) == host.tsv_ctgs([CTGS[0], CTGS[2]])
    with pytest.raises(host.HostError):
        host.bincode_ctg_bundle_decode(struct.pack("<Q", 2**60) + a)
    with pytest.raises(host.HostError):
        host.bincode_ctg_bundle_decode(struct.pack("<QQ", 1, 2**40) + b"xx")


@pytest.mark.parametrize("raw,ok", [(b"ctg:\xce\xb1:1", True), (b"ctg:\xe2\x82\xac:1", True), (b"ctg:\xf0\x9f\xa7\xac:1", True),
                                    (b"ctg:\xff:1", False), (b"ctg:\xc0\xaf:1", False), (b"ctg:\xed\xa0\x80:1", False),
                                    (b"ctg:\xf4\x90\x80\x80:1", False), (b"ctg:\xe2\x82", False), (b"ctg:\x80:1", False)])
def test_bincode_strings_must_be_utf8(raw, ok):
    """a Rust String: bincode 1.3.3 deserialises it through str::from_utf8 (shortest forms, no surrogates,
    <= U+10FFFF) and fails with "invalid utf-8 encoding" otherwise; the decoders here do the same."""
    blob = struct.pack("<Q", 1) + struct.pack("<II", 5, 9) + struct.pack("<Q", len(raw)) + raw
    blob += struct.pack("<QI", 1, 5) + struct.pack("<QI", 1, 9) + struct.pack("<I", 4) + b"\x00\x00"
    if ok:
        assert raw.decode() in host.bincode_lapper_decode(blob)
    else:
        with pytest.raises(host.HostError):
            host.bincode_lapper_decode(blob)


def lapper_bytes(ivs):
    ivs = sorted(ivs, key=lambda v: (v[0], v[1]))                                # Lapper::new: intervals.sort(), stable
    out = struct.pack("<Q", len(ivs)) + b"".join(struct.pack("<II", a, b) + s(v) for a, b, v in ivs)
    out += struct.pack("<Q", len(ivs)) + b"".join(struct.pack("<I", a) for a in sorted(v[0] for v in ivs))
    out += struct.pack("<Q", len(ivs)) + b"".join(struct.pack("<I", b) for b in sorted(v[1] for v in ivs))
    out += struct.pack("<I", max([b - a for a, b, _ in ivs if b > a], default=0))
    return out + b"\x00\x00"                                                     # cov: None, overlaps_merged: false


def test_lapper_idx_ctg_bytes():
    # build_idx_ctg (redis.rs:236-258): Interval{start: chr_start, stop: chr_end + 1, val: ctg_id}, in get_vec_ctg order
    ivs = [(100001, 230219, "ctg:I:2"), (1, 100001, "ctg:I:1")]
    blob = host.bincode_lapper([v[0] for v in ivs], [v[1] for v in ivs], [v[2] for v in ivs])
    assert blob == lapper_bytes(ivs)
    text = host.bincode_lapper_decode(blob)
    assert text == ("1\t100001\tctg:I:1\n100001\t230219\tctg:I:2\n#starts 1 100001\n#stops 100001 230219\n"
                    "#max_len 130218 cov None merged false\n")


def test_lapper_idx_rg_bytes_with_ties_and_nested_intervals():
    # build_idx_rg (redis.rs:276-303): point ranges, val "" -- plus nested / equal intervals for the sort rule
    ivs = [(500, 501, ""), (100, 101, ""), (100, 2000, ""), (100, 101, ""), (7, 9, ""), (2000, 2001, "")]
    blob = host.bincode_lapper([v[0] for v in ivs], [v[1] for v in ivs])
    assert blob == lapper_bytes(ivs)
    assert "#max_len 1900 " in host.bincode_lapper_decode(blob)
    empty = host.bincode_lapper([], [])
    assert empty == struct.pack("<QQQ", 0, 0, 0) + struct.pack("<I", 0) + b"\x00\x00"
    assert host.bincode_lapper_decode(empty) == "#starts\n#stops\n#max_len 0 cov None merged false\n"
    with pytest.raises(host.HostError):
        host.bincode_lapper_decode(blob[:-1])
    with pytest.raises(host.HostError):
        host.bincode_lapper_decode(blob[:-2] + b"\x02\x00")                      # Option tag 2
    some = blob[:-2] + b"\x01" + struct.pack("<I", 1234) + b"\x01"               # cov: Some(1234), merged: true
    assert host.bincode_lapper_decode(some).endswith("#max_len 1900 cov 1234 merged true\n")


def test_resp2_commands():
    assert host.resp_command(["GET", "ctg:I:1"]) == b"*2\r\n$3\r\nGET\r\n$7\r\nctg:I:1\r\n"
    assert host.resp_command(["INCR", "cnt:ctg:I"]) == b"*2\r\n$4\r\nINCR\r\n$9\r\ncnt:ctg:I\r\n"
    gz = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 255, 0, 13, 10, 0])              # binary value with NUL and CRLF inside
    assert host.resp_command([b"SET", b"seq:ctg:I:1", gz]) == \
        b"*3\r\n$3\r\nSET\r\n$11\r\nseq:ctg:I:1\r\n$14\r\n" + gz + b"\r\n"
    ev = host.resp_scan_values("feature:ctg:I:1:*")
    assert ev.startswith(b"*5\r\n$4\r\nEVAL\r\n$279\r\n\nlocal cursor = \"0\";\nlocal list = {};\nrepeat\n")
    assert ev.endswith(b"return list;\n\r\n$1\r\n0\r\n$17\r\nfeature:ctg:I:1:*\r\n$4\r\n1000\r\n")


TRANSCRIPT = [                                                                    # server bytes -> flat form
    (b"+OK\r\n", "+OK\n"),
    (b"-NOSCRIPT No matching script. Please use EVAL.\r\n", "-NOSCRIPT No matching script. Please use EVAL.\n"),
    (b":116\r\n", ":116\n"),
    (b":-3\r\n", ":-3\n"),
    (b"$-1\r\n", "_\n"),
    (b"$0\r\n\r\n", "$0 \n"),
    (b"$5\r\nI:1-7\r\n", "$5 I:1-7\n"),
    (b"*0\r\n", "*0\n"),
    (b"*-1\r\n", "_\n"),
    # SCAN 0 MATCH rg:* COUNT 1000 -> [cursor, [keys]]
    (b"*2\r\n$1\r\n0\r\n*2\r\n$12\r\nrg:ctg:I:1:1\r\n$12\r\nrg:ctg:I:1:2\r\n",
     "*2\n$1 0\n*2\n$12 rg:ctg:I:1:1\n$12 rg:ctg:I:1:2\n"),
    (b"*3\r\n:1\r\n$-1\r\n+x\r\n", "*3\n:1\n_\n+x\n"),
]


def test_resp2_replies_and_streaming():
    for raw, flat in TRANSCRIPT:
        got, used = host.resp_parse(raw)
        assert (got, used) == (flat, len(raw)), raw
        got2, used2 = host.resp_parse(raw + b"+NEXT\r\n")                           # stops at the end of the first reply
        assert (got2, used2) == (flat, len(raw))
        for cut in range(len(raw)):                                              # every proper prefix is "not yet"
            assert host.resp_parse(raw[:cut]) == ("", 0), (raw, cut)
    stream = b"".join(r for r, _ in TRANSCRIPT)                                  # a pipeline's replies back to back
    at, seen = 0, []
    while at < len(stream):
        flat, used = host.resp_parse(stream[at:])
        assert used > 0
        seen.append(flat)
        at += used
    assert seen == [f for _, f in TRANSCRIPT]


@pytest.mark.parametrize("bad", [b"?what\r\n", b":12x\r\n", b"$abc\r\n", b"$-2\r\n", b"$3\r\nabcde\r\n", b"*x\r\n"])
def test_resp2_rejects_malformed(bad):
    with pytest.raises(host.HostError):
        host.resp_parse(bad)
