"""ctypes binding of libgams_host.so: the C++ host layer (gams_amd/host/*.cpp) that mirrors the
reference's per-ctg operators (wave/sw proc_ctg, locate, anno) on top of the C ABI."""
import ctypes as C
import os

import numpy as np

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libgams_host.so")
_h = None


class HostError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gams host error {code}: {msg}")
        self.code = code


def load():
    global _h
    if _h is not None:
        return _h
    _lib.load()
    if not os.path.exists(SO_PATH):
        raise ImportError(f"{SO_PATH} is missing: run __graft_entry__.build()")
    L = C.CDLL(SO_PATH)
    sp = C.POINTER(C.c_char_p)
    ip = C.c_void_p
    L.gams_host_last_error.restype = C.c_char_p
    L.gams_host_last_code.restype = C.c_int
    L.gams_host_free.argtypes = [C.c_void_p]
    L.gams_host_wave.restype = C.c_void_p
    L.gams_host_wave.argtypes = [C.c_void_p, C.c_uint32, sp, sp, ip, ip, C.POINTER(C.c_void_p), C.c_int32,
                                 C.c_int32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_int]
    dp = C.POINTER(C.c_double)
    L.gams_host_wave_timed.restype = C.c_void_p
    L.gams_host_wave_timed.argtypes = [C.c_void_p, C.c_uint32, sp, sp, ip, ip, C.POINTER(C.c_void_p), C.c_int32,
                                       C.c_int32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_int, dp,
                                       C.POINTER(C.c_uint64)]
    L.gams_host_wave_gz.restype = C.c_void_p
    L.gams_host_wave_gz.argtypes = [C.c_void_p, C.c_uint32, sp, sp, ip, ip, C.POINTER(C.c_void_p), C.c_void_p, C.c_int32,
                                    C.c_int32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_int, dp,
                                    C.POINTER(C.c_uint64)]
    L.gams_host_decode_gz_many.restype = C.c_int
    L.gams_host_decode_gz_many.argtypes = [C.c_uint32, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p,
                                           C.c_void_p, C.c_uint32]
    L.gams_host_wave_multi.restype = C.c_void_p
    L.gams_host_wave_multi.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, sp, sp, ip, ip,
                                       C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_uint32, C.c_float,
                                       C.c_float, C.c_float, C.c_int, C.c_uint64]
    L.gams_host_sw.restype = C.c_void_p
    L.gams_host_sw.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.c_uint32,
                               sp, ip, ip, C.c_int32, C.c_int32, C.c_int32]
    L.gams_host_locate.restype = C.c_void_p
    L.gams_host_locate.argtypes = [C.c_void_p, C.c_uint32, sp, sp, ip, ip, C.c_char_p, C.c_int, C.c_char_p]
    L.gams_host_find.restype = C.c_void_p
    L.gams_host_find.argtypes = [C.c_void_p, C.c_uint32, sp, sp, ip, ip, C.c_char_p]
    L.gams_host_anno.restype = C.c_void_p
    L.gams_host_anno.argtypes = [C.c_void_p, C.c_uint32, sp, sp, ip, ip, C.c_char_p, C.c_char_p, C.c_int,
                                 C.c_char_p, C.c_uint32, C.c_uint32]
    L.gams_host_locate_seq.restype = C.c_void_p
    L.gams_host_locate_seq.argtypes = [C.c_void_p, C.c_uint32, sp, sp, ip, ip, C.c_char_p, C.c_char_p]
    L.gams_host_read_range.restype = C.c_void_p
    L.gams_host_read_range.argtypes = [C.c_void_p, C.c_uint32, sp, sp, ip, ip, C.c_char_p]
    L.gams_host_decode_gz.restype = C.c_void_p
    L.gams_host_decode_gz.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.gams_host_encode_gz.restype = C.c_void_p
    L.gams_host_encode_gz.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.gams_host_loader_records.restype = C.c_void_p
    L.gams_host_loader_records.argtypes = [C.c_void_p, C.c_uint32, sp, sp, ip, ip, C.c_char_p, C.c_char_p]
    L.gams_host_count_multi.restype = C.c_int
    L.gams_host_count_multi.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6 + [C.c_uint64, C.c_void_p]
    L.gams_host_cover_multi.restype = C.c_int
    L.gams_host_cover_multi.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 8 + [C.c_uint64, C.c_void_p]
    L.gams_host_last_operator_ms.restype = C.c_double
    L.gams_host_last_operator_ms.argtypes = []
    L.gams_host_sw_multi_timed.restype = C.c_void_p
    L.gams_host_sw_multi_timed.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, sp, sp, ip, ip, C.c_void_p, C.c_char_p,
                                           C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.gams_host_sw_multi.restype = C.c_void_p
    L.gams_host_sw_multi.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, sp, sp, ip, ip, C.c_void_p, C.c_char_p,
                                     C.c_int32, C.c_int32, C.c_int32]
    u64p = C.POINTER(C.c_uint64)
    L.gams_host_bincode_ctg_bundle.restype = C.c_void_p
    L.gams_host_bincode_ctg_bundle.argtypes = [C.c_uint32, sp, sp, ip, ip, u64p]
    L.gams_host_bincode_ctg_bundle_decode.restype = C.c_void_p
    L.gams_host_bincode_ctg_bundle_decode.argtypes = [C.c_void_p, C.c_uint64]
    L.gams_host_bincode_lapper.restype = C.c_void_p
    L.gams_host_bincode_lapper.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, sp, u64p]
    L.gams_host_bincode_lapper_decode.restype = C.c_void_p
    L.gams_host_bincode_lapper_decode.argtypes = [C.c_void_p, C.c_uint64]
    L.gams_host_resp_command.restype = C.c_void_p
    L.gams_host_resp_command.argtypes = [C.c_uint32, sp, u64p, u64p]
    L.gams_host_resp_scan_values.restype = C.c_void_p
    L.gams_host_resp_scan_values.argtypes = [C.c_char_p, u64p]
    L.gams_host_resp_parse.restype = C.c_void_p
    L.gams_host_resp_parse.argtypes = [C.c_char_p, C.c_uint64, u64p]
    L.gams_host_index_from_lappers.restype = C.c_void_p
    L.gams_host_index_from_lappers.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), u64p]
    L.gams_host_header.restype = C.c_void_p
    L.gams_host_header.argtypes = [C.c_int]
    L.gams_host_tsv_ctgs.restype = C.c_void_p
    L.gams_host_tsv_ctgs.argtypes = [C.c_uint32, sp, sp, ip, ip]
    L.gams_host_loader_tsv.restype = C.c_void_p
    L.gams_host_loader_tsv.argtypes = [C.c_void_p, C.c_uint32, sp, sp, ip, ip, C.c_char_p, C.c_char_p]
    L.gams_host_peak.restype = C.c_void_p
    L.gams_host_peak.argtypes = [C.c_void_p, C.c_uint32, sp, sp, ip, ip, C.POINTER(C.c_void_p), C.c_char_p]
    L.gams_host_gen.restype = C.c_void_p
    L.gams_host_gen.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32]
    L.gams_host_fmt_f32.restype = C.c_void_p
    L.gams_host_fmt_f32.argtypes = [C.c_float]
    L.gams_host_range_roundtrip.restype = C.c_void_p
    L.gams_host_range_roundtrip.argtypes = [C.c_char_p]
    _h = L
    return L


def _take(p):
    L = load()
    if not p:
        raise HostError(L.gams_host_last_code(), L.gams_host_last_error().decode(errors="replace"))
    s = C.string_at(p).decode()
    L.gams_host_free(p)
    return s


def _ctg_arrays(ctgs):
    n = len(ctgs)
    ids = (C.c_char_p * max(n, 1))(*[c["id"].encode() for c in ctgs])
    chrs = (C.c_char_p * max(n, 1))(*[c["chr_id"].encode() for c in ctgs])
    st = np.array([c["chr_start"] for c in ctgs], np.int32)
    en = np.array([c["chr_end"] for c in ctgs], np.int32)
    return n, ids, chrs, st, en


def wave(eng, ctgs, size=100, step=10, lag=100, threshold=3.0, influence=1.0, coverage=0.2, is_signal=False):
    """TSV rows (no header) of `gams wave` over `ctgs` (dicts with id, chr_id, chr_start, chr_end, seq)."""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    bufs = [np.ascontiguousarray(np.frombuffer(c["seq"], np.uint8) if not isinstance(c["seq"], np.ndarray)
                                 else c["seq"]) for c in ctgs]
    seqs = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
    return _take(load().gams_host_wave(eng.h, n, ids, chrs, st.ctypes.data, en.ctypes.data, seqs, size, step, lag,
                                       threshold, influence, coverage, int(is_signal)))


STAGE_NAMES = ("inflate_upload_ms", "upload_ms", "plan_ms", "kernel_ms", "peaks_ms", "format_ms", "total_ms",
               "inflate_threads", "peaks")


def _take_bytes(p, n):
    L = load()
    if not p:
        raise HostError(L.gams_host_last_code(), L.gams_host_last_error().decode(errors="replace"))
    b = C.string_at(p, n.value)
    L.gams_host_free(p)
    return b


def _stage_dict(st):
    d = {k: float(v) for k, v in zip(STAGE_NAMES, st)}
    d["inflate_threads"] = int(d["inflate_threads"])
    d["peaks"] = int(d["peaks"])
    return d


def wave_timed(eng, ctgs, size=100, step=10, lag=100, threshold=3.0, influence=1.0, coverage=0.2, sync=False, is_signal=False):
    """`wave` from gunzipped host buffers -> (TSV bytes, stage clock dict of gams::WaveStages)."""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    bufs = [np.ascontiguousarray(np.frombuffer(c["seq"], np.uint8) if not isinstance(c["seq"], np.ndarray)
                                 else c["seq"]) for c in ctgs]
    seqs = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
    stages = (C.c_double * 10)()
    ln = C.c_uint64()
    p = load().gams_host_wave_timed(eng.h, n, ids, chrs, st.ctypes.data, en.ctypes.data, seqs, size, step, lag,
                                    threshold, influence, coverage, int(sync) | (2 if is_signal else 0), stages, C.byref(ln))
    return _take_bytes(p, ln), _stage_dict(stages)


def wave_gz(eng, ctgs, size=100, step=10, lag=100, threshold=3.0, influence=1.0, coverage=0.2, threads=0, sync=False):
    """`wave` from the gzip'd `seq:` values (ctg dicts with id, chr_id, chr_start, chr_end, gz = bytes): `threads`
    host threads inflate one ctg at a time each into a page-locked image of the device buffer (0: 16)
    -> (TSV bytes, stage clock dict)."""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    bufs = [np.frombuffer(c["gz"], np.uint8) for c in ctgs]
    blobs = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
    lens = np.array([b.size for b in bufs], np.uint64)
    stages = (C.c_double * 10)()
    ln = C.c_uint64()
    p = load().gams_host_wave_gz(eng.h, n, ids, chrs, st.ctypes.data, en.ctypes.data, blobs, lens.ctypes.data, size, step,
                                 lag, threshold, influence, coverage, threads, int(sync), stages, C.byref(ln))
    return _take_bytes(p, ln), _stage_dict(stages)


def decode_gz_many(blobs, sizes, threads=0):
    """decode_gz of a batch of `seq:` values on `threads` host threads (0: 16); sizes[i] = room for value i
    -> list of uint8 arrays"""
    L = load()
    n = len(blobs)
    src = [np.frombuffer(b, np.uint8) for b in blobs]
    dst = [np.empty(max(int(s), 1), np.uint8) for s in sizes]
    sp_ = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in src])
    dp_ = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in dst])
    sl = np.array([b.size for b in src], np.uint64)
    cap = np.array([int(s) for s in sizes], np.uint64)
    got = np.zeros(max(n, 1), np.uint64)
    if L.gams_host_decode_gz_many(n, sp_, sl.ctypes.data, dp_, cap.ctypes.data, got.ctypes.data, threads) != 0:
        raise HostError(L.gams_host_last_code(), L.gams_host_last_error().decode(errors="replace"))
    return [d[:int(g)] for d, g in zip(dst, got)]


def wave_multi(engines, ctgs, size=100, step=10, lag=100, threshold=3.0, influence=1.0, coverage=0.2,
               is_signal=False, batch_bytes=1 << 30):
    """`gams wave` over several handles (one per GPU): LPT-sharded ctgs, one host thread per handle."""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    bufs = [np.ascontiguousarray(np.frombuffer(c["seq"], np.uint8) if not isinstance(c["seq"], np.ndarray)
                                 else c["seq"]) for c in ctgs]
    seqs = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
    hs = (C.c_void_p * len(engines))(*[e.h.value for e in engines])
    return _take(load().gams_host_wave_multi(hs, len(engines), n, ids, chrs, st.ctypes.data, en.ctypes.data, seqs,
                                             size, step, lag, threshold, influence, coverage, int(is_signal),
                                             batch_bytes))


def sw(eng, ctg, features, size=100, mx=20, resize=500):
    """features: list of (feature_id, start, end)."""
    nf = len(features)
    fid = (C.c_char_p * max(nf, 1))(*[f[0].encode() for f in features])
    fs = np.array([f[1] for f in features], np.int32)
    fe = np.array([f[2] for f in features], np.int32)
    seq = np.ascontiguousarray(np.frombuffer(ctg["seq"], np.uint8) if not isinstance(ctg["seq"], np.ndarray)
                               else ctg["seq"])
    return _take(load().gams_host_sw(eng.h, ctg["id"].encode(), ctg["chr_id"].encode(), ctg["chr_start"],
                                     ctg["chr_end"], seq.ctypes.data, nf, fid, fs.ctypes.data, fe.ctypes.data,
                                     size, mx, resize))


def locate(eng, ctgs, rgs, count=False, rg_records=()):
    """rgs: list of range strings; rg_records: list of (ctg_id, range string) for --count."""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    rg_lines = "\n".join(f"{c}\t{r}" for c, r in rg_records)
    return _take(load().gams_host_locate(eng.h, n, ids, chrs, st.ctypes.data, en.ctypes.data,
                                         "\n".join(rgs).encode(), int(count), rg_lines.encode()))


def find(eng, ctgs, rgs):
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    out = _take(load().gams_host_find(eng.h, n, ids, chrs, st.ctypes.data, en.ctypes.data, "\n".join(rgs).encode()))
    return out.split("\n")[:len(rgs)]


def anno(eng, ctgs, runlists, lines, header=False, prefix="", idx_id=1, idx_range=2):
    """runlists: dict chr -> runlist string."""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    rl = "\n".join(f"{k}\t{v}" for k, v in runlists.items())
    return _take(load().gams_host_anno(eng.h, n, ids, chrs, st.ctypes.data, en.ctypes.data, rl.encode(),
                                       "\n".join(lines).encode(), int(header), prefix.encode(), idx_id, idx_range))


def gen(eng, chr_id, seq, piece=500000, fill=50, min_len=5000):
    """ctg rows (id, range, chr_id, chr_start, chr_end, chr_strand, length) of one chromosome."""
    a = np.ascontiguousarray(np.frombuffer(seq, np.uint8) if not isinstance(seq, np.ndarray) else seq)
    return _take(load().gams_host_gen(eng.h, chr_id.encode(), a.ctypes.data, a.size, piece, fill, min_len))


def locate_seq(eng, ctgs, rgs):
    """`gams locate --seq`: FASTA text; ctgs carry their gunzipped sequence in c["seq"]."""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    seq_lines = "\n".join(f"{c['id']}\t{bytes(c['seq']).decode()}" for c in ctgs)
    return _take(load().gams_host_locate_seq(eng.h, n, ids, chrs, st.ctypes.data, en.ctypes.data,
                                             "\n".join(rgs).encode(), seq_lines.encode()))


def read_range(eng, ctgs, lines):
    """utils.rs:39-67 incl. the drop-first-per-ctg quirk: list of (ctg_id, range string)."""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    out = _take(load().gams_host_read_range(eng.h, n, ids, chrs, st.ctypes.data, en.ctypes.data,
                                            "\n".join(lines).encode()))
    return [tuple(r.split("\t")) for r in out.splitlines()]


def decode_gz(blob):
    n = C.c_uint64()
    p = load().gams_host_decode_gz(blob, len(blob), C.byref(n))
    if not p:
        raise HostError(load().gams_host_last_code(), load().gams_host_last_error().decode(errors="replace"))
    out = C.string_at(p, n.value)
    load().gams_host_free(p)
    return out


def encode_gz(data):
    n = C.c_uint64()
    p = load().gams_host_encode_gz(data, len(data), C.byref(n))
    if not p:
        raise HostError(-1, "encode_gz failed")
    out = C.string_at(p, n.value)
    load().gams_host_free(p)
    return out


def loader_records(eng, ctgs, lines, tag=None):
    """(key, json) of the rg: (tag None) or feature: records `gams rg` / `gams feature` would SET."""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    out = _take(load().gams_host_loader_records(eng.h, n, ids, chrs, st.ctypes.data, en.ctypes.data,
                                                "\n".join(lines).encode(), tag.encode() if tag is not None else None))
    return [tuple(r.split("\t", 1)) for r in out.splitlines()]


def sw_multi(engines, ctgs, features_per_ctg, size=100, mx=20, resize=500):
    """`gams sw` over several handles; features_per_ctg[i] = list of (id, start, end) of ctgs[i]."""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    bufs = [np.ascontiguousarray(np.frombuffer(c["seq"], np.uint8) if not isinstance(c["seq"], np.ndarray)
                                 else c["seq"]) for c in ctgs]
    seqs = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
    hs = (C.c_void_p * len(engines))(*[e.h.value for e in engines])
    rows = "\n".join(f"{i}\t{fid}\t{s}\t{e}" for i, fl in enumerate(features_per_ctg) for fid, s, e in fl)
    return _take(load().gams_host_sw_multi(hs, len(engines), n, ids, chrs, st.ctypes.data, en.ctypes.data, seqs,
                                           rows.encode(), size, mx, resize))


def last_operator_ms():
    """ms the last locate / anno call of this thread spent inside the C++ operator (without the binding's copies)"""
    return float(load().gams_host_last_operator_ms())


def sw_multi_timed(engines, ctgs, features_per_ctg, size=100, mx=20, resize=500):
    """sw_multi, returning (text, ms of the operator itself: upload + kernels + row text, without this binding's
    parsing and copies)."""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    bufs = [np.ascontiguousarray(np.frombuffer(c["seq"], np.uint8) if not isinstance(c["seq"], np.ndarray)
                                 else c["seq"]) for c in ctgs]
    seqs = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
    hs = (C.c_void_p * len(engines))(*[e.h.value for e in engines])
    rows = "\n".join(f"{i}\t{fid}\t{s}\t{e}" for i, fl in enumerate(features_per_ctg) for fid, s, e in fl)
    ms = C.c_double(0.0)
    n_out = C.c_uint64(0)
    p = load().gams_host_sw_multi_timed(hs, len(engines), n, ids, chrs, st.ctypes.data, en.ctypes.data, seqs,
                                        rows.encode(), size, mx, resize, C.byref(ms), C.byref(n_out))
    return _take_bytes(p, n_out).decode(), ms.value


def _check_rc(rc):
    if rc != 0:
        L = load()
        raise HostError(L.gams_host_last_code(), L.gams_host_last_error().decode(errors="replace"))


def count_multi(engines, group_off, starts, stops, q_group, qs, qe):
    """`locate --count` over several handles: groups split by LPT, queries routed to their group's device."""
    hs = (C.c_void_p * len(engines))(*[e.h.value for e in engines])
    group_off = np.ascontiguousarray(group_off, np.uint64)
    a = [np.ascontiguousarray(x, np.uint32) for x in (starts, stops, q_group, qs, qe)]
    out = np.zeros(a[2].size, np.int32)
    _check_rc(load().gams_host_count_multi(hs, len(engines), group_off.size - 1, group_off.ctypes.data,
                                           *[x.ctypes.data for x in a], a[2].size, out.ctypes.data))
    return out


def cover_multi(engines, group_off, lo, hi, q_group, clip_lo, clip_hi, qs, qe):
    """`anno` coverage over several handles (groups = chromosomes of the runlist set)."""
    hs = (C.c_void_p * len(engines))(*[e.h.value for e in engines])
    group_off = np.ascontiguousarray(group_off, np.uint64)
    lo, hi = np.ascontiguousarray(lo, np.int32), np.ascontiguousarray(hi, np.int32)
    q_group = np.ascontiguousarray(q_group, np.uint32)
    b = [np.ascontiguousarray(x, np.int32) for x in (clip_lo, clip_hi, qs, qe)]
    out = np.zeros(q_group.size, np.float32)
    _check_rc(load().gams_host_cover_multi(hs, len(engines), group_off.size - 1, group_off.ctypes.data, lo.ctypes.data,
                                           hi.ctypes.data, q_group.ctypes.data, *[x.ctypes.data for x in b],
                                           q_group.size, out.ctypes.data))
    return out


def merge_windows(windows, chr_start, size, step, coverage):
    """merge_ints (wave.rs:217-252) of ascending window indices: (component min, component max, has-an-edge)"""
    L = load()
    w = np.ascontiguousarray(windows, np.uint32)
    cmin, cmax = np.zeros(w.size, np.int64), np.zeros(w.size, np.int64)
    ing = np.zeros(w.size, np.int8)
    L.gams_host_merge_windows.restype = None
    L.gams_host_merge_windows.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
    L.gams_host_merge_windows(w.ctypes.data, w.size, chr_start, size, step, coverage, cmin.ctypes.data,
                              cmax.ctypes.data, ing.ctypes.data)
    return cmin, cmax, ing.astype(bool)


def ctg_json_roundtrip(json_text):
    """parse a `ctg:` JSON record (redis.rs:132-135) and write it back the way serde_json does (:127-130)"""
    L = load()
    L.gams_host_ctg_json_roundtrip.restype = C.c_void_p
    L.gams_host_ctg_json_roundtrip.argtypes = [C.c_char_p]
    return _take(L.gams_host_ctg_json_roundtrip(json_text.encode()))


def header(command):
    """the header line `gams wave` / `gams sw` print before the rows"""
    return _take(load().gams_host_header(0 if command == "wave" else 1))


def tsv_ctgs(ctgs):
    """`gams tsv -s "ctg:*"` for these ctgs (tsv.rs:31-71)."""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    return _take(load().gams_host_tsv_ctgs(n, ids, chrs, st.ctypes.data, en.ctypes.data))


def loader_tsv(eng, ctgs, lines, tag=None):
    """`gams tsv -s "rg:*"` / `"feature:*"` of what the loaders would have stored."""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    return _take(load().gams_host_loader_tsv(eng.h, n, ids, chrs, st.ctypes.data, en.ctypes.data,
                                             "\n".join(lines).encode(), tag.encode() if tag is not None else None))


def peak(eng, ctgs, lines):
    """`gams peak` over the rows of a wave TSV: TSV of the Peak fields (data.rs:30-43)."""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    bufs = [np.ascontiguousarray(np.frombuffer(c["seq"], np.uint8) if not isinstance(c["seq"], np.ndarray)
                                 else c["seq"]) for c in ctgs]
    seqs = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in bufs])
    return _take(load().gams_host_peak(eng.h, n, ids, chrs, st.ctypes.data, en.ctypes.data, seqs,
                                       "\n".join(lines).encode()))


def fmt_f32(v):
    return _take(load().gams_host_fmt_f32(v))


def range_roundtrip(s):
    return _take(load().gams_host_range_roundtrip(s.encode()))


# ---- wire formats either side of the path (gams_wire.cpp; SURVEY 8 f-3, parity unpinned) --------
def bincode_ctg_bundle(ctgs):
    """bundle:ctg:{chr}: bincode 1.3.3 of BTreeMap<String, Ctg> (redis.rs:216-233)"""
    n, ids, chrs, st, en = _ctg_arrays(ctgs)
    ln = C.c_uint64()
    return _take_bytes(load().gams_host_bincode_ctg_bundle(n, ids, chrs, st.ctypes.data, en.ctypes.data, C.byref(ln)), ln)


def bincode_ctg_bundle_decode(blob):
    """-> the ctg.tsv text of the decoded map"""
    return _take(load().gams_host_bincode_ctg_bundle_decode(blob, len(blob)))


def bincode_lapper(starts, stops, vals=None):
    """idx:ctg:{chr} / idx:rg:{ctg}: Lapper::new(ivs) serialised (redis.rs:236-324)"""
    a = np.ascontiguousarray(starts, np.uint32)
    b = np.ascontiguousarray(stops, np.uint32)
    vp = None
    if vals is not None:
        vp = (C.c_char_p * max(len(vals), 1))(*[v.encode() for v in vals])
    ln = C.c_uint64()
    return _take_bytes(load().gams_host_bincode_lapper(a.size, a.ctypes.data, b.ctypes.data, vp, C.byref(ln)), ln)


def bincode_lapper_decode(blob):
    return _take(load().gams_host_bincode_lapper_decode(blob, len(blob)))


def resp_command(args):
    raw = [a if isinstance(a, bytes) else str(a).encode() for a in args]
    ptrs = (C.c_char_p * max(len(raw), 1))(*raw)
    lens = (C.c_uint64 * max(len(raw), 1))(*[len(a) for a in raw])
    ln = C.c_uint64()
    return _take_bytes(load().gams_host_resp_command(len(raw), ptrs, lens, C.byref(ln)), ln)


def resp_scan_values(pattern):
    ln = C.c_uint64()
    return _take_bytes(load().gams_host_resp_scan_values(pattern.encode(), C.byref(ln)), ln)


def resp_parse(buf):
    """-> (flat text of the first reply in buf, bytes consumed); consumed == 0: the reply is incomplete"""
    used = C.c_uint64()
    p = load().gams_host_resp_parse(buf, len(buf), C.byref(used))
    if not p:
        raise HostError(load().gams_host_last_code(), load().gams_host_last_error().decode(errors="replace"))
    out = C.string_at(p).decode(errors="replace")
    load().gams_host_free(p)
    return out, used.value


def index_from_lappers(eng, blobs):
    """device index (gams_index_t*, as c_void_p) over idx: blobs, one group per blob; free with gams_index_destroy"""
    ptrs = (C.c_char_p * max(len(blobs), 1))(*blobs)
    lens = (C.c_uint64 * max(len(blobs), 1))(*[len(b) for b in blobs])
    p = load().gams_host_index_from_lappers(eng.h, len(blobs), ptrs, lens)
    if not p:
        raise HostError(load().gams_host_last_code(), load().gams_host_last_error().decode(errors="replace"))
    return C.c_void_p(p)
