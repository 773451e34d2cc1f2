"""ctypes view of oracle/libgams_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (gams_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgams_oracle.so")


def build(force=False):
    src = max((os.path.join(_HERE, f) for f in ("gams_oracle.c", "gams_ref.c", "gams_ref.h")), key=os.path.getmtime)
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libgams_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u8p = C.POINTER(C.c_uint8)
        i32p = C.POINTER(C.c_int32)
        u32p = C.POINTER(C.c_uint32)
        f32p = C.POINTER(C.c_float)
        L.ora_sliding_count.restype = C.c_int64
        L.ora_sliding_count.argtypes = [C.c_int64, C.c_int32, C.c_int32]
        L.ora_center_resize.restype = None
        L.ora_center_resize.argtypes = [C.c_int32] * 5 + [i32p, i32p]
        L.ora_center_sw.restype = C.c_int32
        L.ora_center_sw.argtypes = [C.c_int32] * 6 + [i32p] * 4
        L.ora_gc_count.restype = C.c_uint32
        L.ora_gc_count.argtypes = [C.c_void_p, C.c_size_t]
        L.ora_gc_content.restype = C.c_float
        L.ora_gc_content.argtypes = [C.c_void_p, C.c_size_t]
        L.ora_mean.restype = C.c_float
        L.ora_mean.argtypes = [f32p, C.c_size_t]
        L.ora_stddev.restype = C.c_float
        L.ora_stddev.argtypes = [f32p, C.c_size_t]
        L.ora_thresholding_algo.restype = C.c_int
        L.ora_thresholding_algo.argtypes = [f32p, C.c_size_t, C.c_size_t, C.c_float, C.c_float, i32p]
        L.ora_round.restype = C.c_float
        L.ora_round.argtypes = [C.c_float, C.c_uint32]
        L.ora_gc_stat.restype = None
        L.ora_gc_stat.argtypes = [f32p, C.c_size_t, f32p, f32p, f32p]
        L.ora_range_gc_content.restype = C.c_float
        L.ora_range_gc_content.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.ora_range_gc_stat.restype = None
        L.ora_range_gc_stat.argtypes = [C.c_void_p] + [C.c_int32] * 5 + [f32p] * 3
        L.ora_wave_windows.restype = C.c_int64
        L.ora_wave_windows.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_size_t,
                                       C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_wave_proc_ctg.restype = C.c_void_p
        L.ora_wave_proc_ctg.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                        C.c_int32, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_int]
        L.ora_sw_proc_ctg.restype = C.c_void_p
        L.ora_sw_proc_ctg.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_void_p,
                                      C.POINTER(C.c_char_p), i32p, i32p, C.c_size_t,
                                      C.c_int32, C.c_int32, C.c_int32]
        L.ora_lapper_find_first.restype = C.c_int64
        L.ora_lapper_find_first.argtypes = [u32p, u32p, C.c_size_t, C.c_uint32, C.c_uint32]
        L.ora_lapper_count.restype = C.c_int32
        L.ora_lapper_count.argtypes = [u32p, u32p, C.c_size_t, C.c_uint32, C.c_uint32]
        L.ora_anno_prop.restype = C.c_float
        L.ora_anno_prop.argtypes = [i32p, i32p, C.c_size_t] + [C.c_int32] * 4
        L.ora_gen_regions.restype = C.c_int64
        L.ora_gen_regions.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, i32p, i32p, C.c_int64]
        L.ora_peak_rows.restype = C.c_void_p
        L.ora_peak_rows.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, i32p, i32p,
                                    C.POINTER(C.c_char_p), C.c_size_t]
        L.ora_fmt_f32.restype = C.c_int
        L.ora_fmt_f32.argtypes = [C.c_float, C.c_char_p]
        L.ora_free.restype = None
        L.ora_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _u8(buf):
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    return np.ascontiguousarray(a, dtype=np.uint8)


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _take_str(p):
    if not p:
        return None
    s = C.string_at(p).decode()
    lib().ora_free(p)
    return s


def sliding_count(parent_size, size, step):
    return lib().ora_sliding_count(parent_size, size, step)


def center_resize(ps, pe, s, e, resize):
    a, b = C.c_int32(), C.c_int32()
    lib().ora_center_resize(ps, pe, s, e, resize, C.byref(a), C.byref(b))
    return a.value, b.value


def center_sw(ps, pe, start, end, size, mx):
    cap = 1 + 2 * max(mx, 0)
    ws, we, wt, wd = (np.zeros(cap, np.int32) for _ in range(4))
    n = lib().ora_center_sw(ps, pe, start, end, size, mx, _ptr(ws, C.c_int32), _ptr(we, C.c_int32),
                            _ptr(wt, C.c_int32), _ptr(wd, C.c_int32))
    return [(int(ws[i]), int(we[i]), "MLR"[wt[i]], int(wd[i])) for i in range(n)]


def gc_count(seq):
    a = _u8(seq)
    return lib().ora_gc_count(a.ctypes.data, a.size)


def gc_content(seq):
    a = _u8(seq)
    return lib().ora_gc_content(a.ctypes.data, a.size)


def mean(x):
    a = np.ascontiguousarray(x, np.float32)
    return lib().ora_mean(_ptr(a, C.c_float), a.size)


def stddev(x):
    a = np.ascontiguousarray(x, np.float32)
    return lib().ora_stddev(_ptr(a, C.c_float), a.size)


def thresholding_algo(data, lag, threshold, influence):
    a = np.ascontiguousarray(data, np.float32)
    out = np.zeros(a.size, np.int32)
    rc = lib().ora_thresholding_algo(_ptr(a, C.c_float), a.size, lag, threshold, influence,
                                     _ptr(out, C.c_int32))
    if rc != 0:
        raise ValueError("reference panics: n < lag or lag == 0")
    return out


def round_(x, d):
    return lib().ora_round(x, d)


def gc_stat(gcs):
    a = np.ascontiguousarray(gcs, np.float32)
    m, s, c = C.c_float(), C.c_float(), C.c_float()
    lib().ora_gc_stat(_ptr(a, C.c_float), a.size, C.byref(m), C.byref(s), C.byref(c))
    return m.value, s.value, c.value


def range_gc_content(seq, chr_start, rs, re):
    a = _u8(seq)
    return lib().ora_range_gc_content(a.ctypes.data, chr_start, rs, re)


def range_gc_stat(seq, chr_start, rs, re, size, step):
    a = _u8(seq)
    m, s, c = C.c_float(), C.c_float(), C.c_float()
    lib().ora_range_gc_stat(a.ctypes.data, chr_start, rs, re, size, step, C.byref(m), C.byref(s), C.byref(c))
    return m.value, s.value, c.value


def wave_windows(seq, size, step, lag, threshold, influence, want_signals=True):
    """(gc_count u32[n], gc f32[n], signals i32[n]) computed the reference way."""
    a = _u8(seq)
    n = sliding_count(a.size, size, step)
    if n < 0:
        raise ValueError("bad size/step")
    cnt = np.zeros(max(n, 1), np.uint32)
    gc = np.zeros(max(n, 1), np.float32)
    sig = np.zeros(max(n, 1), np.int32)
    rc = lib().ora_wave_windows(a.ctypes.data, a.size, size, step, lag, threshold, influence,
                                cnt.ctypes.data, gc.ctypes.data, sig.ctypes.data if want_signals else None)
    if rc < 0:
        raise ValueError("reference panics: n < lag or lag == 0")
    return cnt[:n], gc[:n], sig[:n]


def wave_proc_ctg(chr_id, chr_start, chr_end, seq, size=100, step=10, lag=100, threshold=3.0,
                  influence=1.0, coverage=0.2, is_signal=False):
    a = _u8(seq)
    assert a.size == chr_end - chr_start + 1
    p = lib().ora_wave_proc_ctg(chr_id.encode(), chr_start, chr_end, a.ctypes.data, size, step, lag,
                                threshold, influence, coverage, int(is_signal))
    s = _take_str(p)
    if s is None:
        raise ValueError("reference panics: n < lag or lag == 0")
    return s


def sw_proc_ctg(chr_id, chr_start, chr_end, seq, features, size=100, mx=20, resize=500):
    """features: list of (feature_id, start, end)."""
    a = _u8(seq)
    nf = len(features)
    ids = (C.c_char_p * max(nf, 1))(*[f[0].encode() for f in features])
    fs = np.array([f[1] for f in features], np.int32)
    fe = np.array([f[2] for f in features], np.int32)
    p = lib().ora_sw_proc_ctg(chr_id.encode(), chr_start, chr_end, a.ctypes.data, ids,
                              _ptr(fs, C.c_int32), _ptr(fe, C.c_int32), nf, size, mx, resize)
    return _take_str(p)


def lapper_find_first(starts, stops, qs, qe):
    s = np.ascontiguousarray(starts, np.uint32)
    t = np.ascontiguousarray(stops, np.uint32)
    return lib().ora_lapper_find_first(_ptr(s, C.c_uint32), _ptr(t, C.c_uint32), s.size, qs, qe)


def lapper_count(sorted_starts, sorted_stops, qs, qe):
    s = np.ascontiguousarray(sorted_starts, np.uint32)
    t = np.ascontiguousarray(sorted_stops, np.uint32)
    return lib().ora_lapper_count(_ptr(s, C.c_uint32), _ptr(t, C.c_uint32), s.size, qs, qe)


def anno_prop(span_lo, span_hi, ctg_s, ctg_e, rs, re):
    lo = np.ascontiguousarray(span_lo, np.int32)
    hi = np.ascontiguousarray(span_hi, np.int32)
    return lib().ora_anno_prop(_ptr(lo, C.c_int32), _ptr(hi, C.c_int32), lo.size, ctg_s, ctg_e, rs, re)


def fmt_f32(v):
    b = C.create_string_buffer(64)
    lib().ora_fmt_f32(v, b)
    return b.value.decode()


def gen_regions(seq, piece=500000, fill=50, min_len=5000):
    """[(start, end)] of the ctgs `gams gen` cuts one chromosome into (gen.rs:81-126)."""
    a = _u8(seq)
    n = lib().ora_gen_regions(a.ctypes.data, a.size, piece, fill, min_len, None, None, 0)
    s = np.zeros(max(n, 1), np.int32)
    e = np.zeros(max(n, 1), np.int32)
    lib().ora_gen_regions(a.ctypes.data, a.size, piece, fill, min_len, _ptr(s, C.c_int32), _ptr(e, C.c_int32), n)
    return [(int(s[i]), int(e[i])) for i in range(n)]


def peak_rows(ctg_id, chr_id, chr_start, chr_end, seq, peaks):
    """peaks: list of (start, end, signal string) in bucket order (peak.rs:65-158)."""
    a = _u8(seq)
    n = len(peaks)
    ps = np.array([p[0] for p in peaks], np.int32)
    pe = np.array([p[1] for p in peaks], np.int32)
    sg = (C.c_char_p * max(n, 1))(*[p[2].encode() for p in peaks])
    return _take_str(lib().ora_peak_rows(ctg_id.encode(), chr_id.encode(), chr_start, chr_end, a.ctypes.data,
                                         _ptr(ps, C.c_int32), _ptr(pe, C.c_int32), sg, n))


# ---- CPU twins of the C ABI (gams_ref.h): the ABI's argument lists over the functions above -----------------
_ref = None


def ref():
    """libgams_oracle.so with the gams_ref_* prototypes bound: argument for argument what gams_amd/_lib.py binds
    for the matching gams_gpu_* entry, without the handle / device objects."""
    global _ref
    if _ref is not None:
        return _ref
    L = lib()
    VP, U64P = C.c_void_p, C.POINTER(C.c_uint64)

    class WaveParams(C.Structure):
        _fields_ = [("size", C.c_int32), ("step", C.c_int32), ("lag", C.c_uint32), ("threshold", C.c_float),
                    ("influence", C.c_float)]

    L.WaveParams = WaveParams
    L.gams_ref_wave.restype = C.c_int
    L.gams_ref_wave.argtypes = [VP, C.c_uint32, C.POINTER(WaveParams), VP, VP, C.POINTER(C.c_uint32)]
    L.gams_ref_wave_peaks.restype = C.c_int
    L.gams_ref_wave_peaks.argtypes = [C.c_uint32, VP, VP, C.POINTER(WaveParams), VP, C.c_uint64, U64P]
    L.gams_ref_wave_rows.restype = C.c_int
    L.gams_ref_wave_rows.argtypes = [C.c_char_p, C.c_int32, VP, C.c_uint32, C.POINTER(WaveParams), C.c_float,
                                     C.POINTER(C.c_void_p), U64P]
    L.gams_ref_wave_signal_text.restype = C.c_int
    L.gams_ref_wave_signal_text.argtypes = [C.c_char_p, C.c_int32, VP, C.c_uint32, C.POINTER(WaveParams), C.POINTER(C.c_void_p), U64P]
    L.gams_ref_sw_text.restype = C.c_int
    L.gams_ref_sw_text.argtypes = [C.c_char_p, VP, C.c_uint32, C.c_int32, VP, VP, C.POINTER(C.c_char_p), C.c_uint32, C.c_int32,
                                   C.c_int32, C.c_int32, C.POINTER(C.c_void_p), U64P]
    L.gams_ref_sw.restype = C.c_int
    L.gams_ref_sw.argtypes = [VP, C.c_uint32, C.c_int32, VP, VP, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, VP,
                              C.c_uint64, U64P]
    L.gams_ref_range_gc.restype = C.c_int
    L.gams_ref_range_gc.argtypes = [VP, C.c_uint32, C.c_int32, VP, VP, C.c_uint32, VP]
    L.gams_ref_count.restype = C.c_int
    L.gams_ref_count.argtypes = [C.c_uint32, VP, VP, VP, VP, VP, VP, C.c_uint64, VP]
    L.gams_ref_locate.restype = C.c_int
    L.gams_ref_locate.argtypes = [C.c_uint32, VP, VP, VP, VP, VP, VP, C.c_uint64, VP]
    L.gams_ref_cover.restype = C.c_int
    L.gams_ref_cover.argtypes = [C.c_uint32, VP, VP, VP, VP, VP, VP, VP, VP, C.c_uint64, VP]
    L.gams_ref_valid_spans.restype = C.c_int
    L.gams_ref_valid_spans.argtypes = [VP, C.c_uint64, C.c_int32, C.c_int32, VP, VP, C.c_uint64, U64P]
    L.gams_ref_free.restype = None
    L.gams_ref_free.argtypes = [VP]
    _ref = L
    return L
