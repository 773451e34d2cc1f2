"""Shared test helpers: fixture loading and a tiny restatement of the host-side
bookkeeping around the hot path (ctg split, range parsing, rg/feature bucketing).

Citations are file:line under the reference (wang-q/gams).  The arithmetic the
tests check lives in oracle/ (CPU) and gams_amd/csrc (HIP); what is here is only
the glue the reference's own CLI tests go through before they reach it.
"""
import gzip
import json
import os
import re

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
S288C = os.path.join(GOLDEN, "S288c")


def read_fasta_gz(path):
    seqs, name, chunks = {}, None, []
    with gzip.open(path, "rb") as fh:
        for line in fh:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    seqs[name] = b"".join(chunks)
                name = line[1:].split()[0].decode()
                chunks = []
            else:
                chunks.append(line)
    if name is not None:
        seqs[name] = b"".join(chunks)
    return seqs


def gen_ctgs(chr_id, seq, piece=500000, fill=50, min_len=5000):
    """cmd_gams/gen.rs:81-157: valid regions -> --piece chunks -> ctg records."""
    a = np.frombuffer(seq, np.uint8)
    ok = np.isin(a, np.frombuffer(b"ACGTacgt", np.uint8))            # gen.rs:86-93
    # valid spans (1-based inclusive)
    d = np.diff(np.concatenate(([0], ok.view(np.int8), [0])))
    starts = np.flatnonzero(d == 1) + 1
    ends = np.flatnonzero(d == -1)
    spans = [[int(s), int(e)] for s, e in zip(starts, ends)]
    # fill(fill-1): holes of size <= fill-1 are closed (gen.rs:103)
    filled = []
    for s, e in spans:
        if filled and s - filled[-1][1] - 1 <= fill - 1:
            filled[-1][1] = e
        else:
            filled.append([s, e])
    # excise(min): spans shorter than min are dropped (gen.rs:104)
    filled = [sp for sp in filled if sp[1] - sp[0] + 1 >= min_len]
    ctgs = []
    serial = 0
    for pos, mx in filled:                                            # gen.rs:108-126
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
                             range=f"{chr_id}:{s}-{e}", length=e - s + 1, seq=seq[s - 1:e]))
    return ctgs


_RG = re.compile(r"^(?:(?P<name>[\w_]+)\.)?(?P<chr>[\w-]+)(?:\((?P<strand>[+-])\))?"
                 r"(?::(?P<start>\d+)(?:[_\-]+(?P<end>\d+))?)?$")


def parse_range(s):
    """intspan::Range::from_str: chr(strand):start-end; returns None if invalid."""
    m = _RG.match(s.strip())
    if not m or m.group("start") is None:
        return None
    start = int(m.group("start"))
    end = int(m.group("end")) if m.group("end") else start
    return m.group("chr"), start, end


def ctg_index(ctgs):
    """redis.rs:236-258: per chr, intervals (chr_start, chr_end+1, ctg_id) sorted."""
    idx = {}
    for c in ctgs:
        idx.setdefault(c["chr_id"], []).append((c["chr_start"], c["chr_end"] + 1, c["id"]))
    for k in idx:
        idx[k].sort()
    return idx


def load_s288c():
    seqs = read_fasta_gz(os.path.join(S288C, "genome.fa.gz"))
    return seqs


def read_lines(name):
    with open(os.path.join(S288C, name)) as fh:
        return [ln.rstrip("\n") for ln in fh]


def read_runlists(name):
    with open(os.path.join(S288C, name)) as fh:
        js = json.load(fh)
    out = {}
    for chr_id, rl in js.items():
        lo, hi = [], []
        for part in rl.split(","):
            if part in ("", "-"):
                continue
            if "-" in part:
                a, b = part.split("-")
            else:
                a = b = part
            lo.append(int(a))
            hi.append(int(b))
        out[chr_id] = (np.array(lo, np.int32), np.array(hi, np.int32))
    return out
