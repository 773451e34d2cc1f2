#!/usr/bin/env python3
"""Mutation fuzz of the byte decoders of the host layer (gams::wire: bincode bundle:ctg: / idx:* blobs, RESP2
replies, gzip members, ctg JSON, range strings): every mutated input must either decode or raise HostError -- never crash,
hang or read out of bounds.  Meant to be run against an AddressSanitizer build of libgams_host.so on the CPU:

    g++ -O1 -g -fsanitize=address,undefined ... -o gams_amd/libgams_host.so gams_amd/host/*.cpp ...
    LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \\
        ASAN_OPTIONS=detect_leaks=0 python tools/fuzz_wire.py 20000

usage: tools/fuzz_wire.py [rounds] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gams_amd import host  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)

ctgs = [dict(id=f"ctg:I:{k + 1}", chr_id="I", chr_start=1 + 1000 * k, chr_end=1000 * (k + 1)) for k in range(5)]
seeds = {
    "bundle": [host.bincode_ctg_bundle(ctgs), host.bincode_ctg_bundle(ctgs[:1]), host.bincode_ctg_bundle([])],
    "lapper": [host.bincode_lapper([1, 5, 9], [4, 8, 20], ["a", "bb", ""]), host.bincode_lapper([], []),
               host.bincode_lapper(np.arange(0, 4000, 4), np.arange(3, 4003, 4))],
    "resp": [b"+OK\r\n", b"-ERR x\r\n", b":42\r\n", b"$5\r\nhello\r\n", b"$-1\r\n", b"*2\r\n$1\r\na\r\n:7\r\n",
             b"*-1\r\n", b"*1\r\n*1\r\n*1\r\n+deep\r\n", host.resp_command(["SET", "k", b"\x00\xff" * 10])],
    "gz": [host.encode_gz(b"ACGT" * 500), host.encode_gz(b"")],
    "json": [b'{"id":"ctg:I:1","range":"I:1-100","chr_id":"I","chr_start":1,"chr_end":100,"chr_strand":"+","length":100}'],
    "range": [b"I:1-100", b"I(+):1-100", b"Mito(-):5", b"chr1:1,000-2,000", b"I", b"scaffold_12:7-7", b"X:-5-10"],
}
decode = {
    "bundle": host.bincode_ctg_bundle_decode,
    "lapper": host.bincode_lapper_decode,
    "resp": host.resp_parse,
    "gz": host.decode_gz,
    "json": lambda b: host.ctg_json_roundtrip(b.decode(errors="replace")),
    "range": lambda b: host.range_roundtrip(b.replace(b"\0", b"0").decode(errors="replace")),
}


def mutate(b):
    a = bytearray(b)
    kind = rng.integers(0, 7)
    if kind == 0 and a:                                  # flip a few bits
        for _ in range(int(rng.integers(1, 4))):
            a[int(rng.integers(0, len(a)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1 and a:                                # truncate
        del a[int(rng.integers(0, len(a))):]
    elif kind == 2 and a:                                # overwrite a run with random bytes
        i = int(rng.integers(0, len(a)))
        n = int(rng.integers(1, 9))
        a[i:i + n] = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    elif kind == 3 and len(a) >= 8:                      # a huge length / count field somewhere
        i = int(rng.integers(0, len(a) - 7))
        a[i:i + 8] = [2**63, 2**64 - 1, 2**32, 2**31 - 1, 10**12][int(rng.integers(0, 5))].to_bytes(8, "little")
    elif kind == 4:                                      # append garbage
        a += rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
    elif kind == 5 and a:                                # delete a slice from the middle
        i = int(rng.integers(0, len(a)))
        del a[i:i + int(rng.integers(1, 16))]
    else:                                                # pure noise
        a = bytearray(rng.integers(0, 256, int(rng.integers(0, 64)), dtype=np.uint8).tobytes())
    return bytes(a)


ok = bad = 0
for r in range(rounds):
    for name, blobs in seeds.items():
        blob = mutate(blobs[int(rng.integers(0, len(blobs)))])
        if rng.random() < 0.3:
            blob = mutate(blob)
        try:
            decode[name](blob)
            ok += 1
        except host.HostError:
            bad += 1
print(f"wire fuzz: {rounds} rounds x {len(seeds)} decoders: {ok} inputs decoded, {bad} rejected with an error, no crash")
