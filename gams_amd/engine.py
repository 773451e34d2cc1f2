"""Thin Python objects over the C ABI (include/gams_gpu.h) for tests and bench.py.

Plumbing only: every computation happens in libgams_gpu.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import GamsError, WaveParams, PEAK_DTYPE, SW_ROW_DTYPE, WAVE_PEAKS, WAVE_DENSE


def _u8(buf):
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf, dtype=np.uint8)
    return np.frombuffer(buf, dtype=np.uint8)


class Engine:
    """One handle = one device + its streams (gams_gpu_create)."""

    def __init__(self, device=0, lib=None):
        self.lib = lib if lib is not None else _lib.load()   # `lib`: an alternative build (tools/ab.py)
        h = C.c_void_p()
        rc = self.lib.gams_gpu_create(device, C.byref(h))
        if rc != _lib.OK:
            raise GamsError(rc, "gams_gpu_create failed (no HIP device?)")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.gams_gpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != _lib.OK:
            raise GamsError(rc, self.lib.gams_gpu_last_error(self.h).decode(errors="replace"))

    def device_info(self):
        arch = C.create_string_buffer(64)
        cus, hbm = C.c_int32(), C.c_uint64()
        self.check(self.lib.gams_gpu_device_info(self.h, arch, 64, C.byref(cus), C.byref(hbm)))
        return arch.value.decode(), cus.value, hbm.value

    def sync(self):
        self.check(self.lib.gams_gpu_sync(self.h))

    def timer_start(self):
        self.check(self.lib.gams_gpu_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self.check(self.lib.gams_gpu_timer_stop(self.h, C.byref(ms)))
        return ms.value

    # ---- one-shot wave over a host buffer (wave.rs:143-155) ----
    def wave(self, seq, size=100, step=10, lag=100, threshold=3.0, influence=1.0):
        a = _u8(seq)
        n = self.lib.gams_window_count(a.size, size, step)
        if n < 0:
            raise GamsError(_lib.EINVAL, "bad size/step")
        cnt = np.zeros(max(n, 1), np.uint32)
        sig = np.zeros(max(n, 1), np.int8)
        prm = WaveParams(size, step, lag, threshold, influence)
        nw = C.c_uint32()
        self.check(self.lib.gams_gpu_wave(self.h, a.ctypes.data, a.size, C.byref(prm), cnt.ctypes.data,
                                          sig.ctypes.data, C.byref(nw)))
        return cnt[:nw.value], sig[:nw.value]


class SeqSet:
    """HBM image of a batch of ctg sequences (gams_seqset_*)."""

    def __init__(self, eng, seqs):
        self.eng = eng
        self.lengths = np.array([len(s) for s in seqs], np.uint32)
        p = C.c_void_p()
        eng.check(eng.lib.gams_seqset_create(eng.h, len(seqs), self.lengths.ctypes.data, C.byref(p)))
        self.p = p
        arrs = [_u8(s) for s in seqs]
        ptrs = (C.c_void_p * max(len(arrs), 1))(*[a.ctypes.data if a.size else None for a in arrs])
        eng.check(eng.lib.gams_seqset_upload_all(eng.h, self.p, ptrs))

    def upload(self, i, seq):
        """replace ctg i (same length) with new bases"""
        a = _u8(seq)
        if a.size != int(self.lengths[i]):
            raise ValueError("SeqSet.upload: length differs from the ctg's slot")
        self.eng.check(self.eng.lib.gams_seqset_upload(self.eng.h, self.p, i, a.ctypes.data))

    def close(self):
        if getattr(self, "p", None):
            self.eng.lib.gams_seqset_destroy(self.eng.h, self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class WavePlan:
    """Geometry + device buffers of one `wave` configuration over a SeqSet."""

    def __init__(self, eng, seqset, size=100, step=10, lag=100, threshold=3.0, influence=1.0,
                 flags=WAVE_PEAKS, tile_windows=0):
        self.eng, self.seqset = eng, seqset
        self.prm = WaveParams(size, step, lag, threshold, influence)
        p = C.c_void_p()
        eng.check(eng.lib.gams_wave_plan_create(eng.h, seqset.p, C.byref(self.prm), flags, C.byref(p)))
        self.p = p
        self.flags = flags
        if tile_windows:
            eng.check(eng.lib.gams_wave_plan_set_tile(eng.h, self.p, tile_windows))

    @property
    def total_windows(self):
        return self.eng.lib.gams_wave_total_windows(self.p)

    def ctg_windows(self, i):
        return self.eng.lib.gams_wave_ctg_windows(self.p, i)

    def set_depth(self, depth):
        """passes in flight: consecutive run() calls rotate over `depth` streams and output sets"""
        self.eng.check(self.eng.lib.gams_wave_plan_set_depth(self.eng.h, self.p, depth))

    def set_tile(self, tile_windows):
        """windows per workgroup (0: the library's choice); 1024 / 2048 / 3072 / 5120 / 7168 select the fast kernels' W"""
        self.eng.check(self.eng.lib.gams_wave_plan_set_tile(self.eng.h, self.p, tile_windows))

    def settled(self):
        """influence != 1: (sweeps of guess-and-iterate the selected pass took, handed to the serial recurrence?)"""
        n, ser = C.c_uint32(), C.c_int()
        self.eng.check(self.eng.lib.gams_wave_plan_settled(self.eng.h, self.p, C.byref(n), C.byref(ser)))
        return int(n.value), bool(ser.value)

    def set_threads(self, threads):
        """threads per workgroup of the step-1 W = 28 kernels: 64 / 128 / 256 (0: the library's choice)"""
        self.eng.check(self.eng.lib.gams_wave_plan_set_threads(self.eng.h, self.p, threads))

    def set_lane(self, lane):
        """run this plan on HIP stream `lane` (+ way) of the handle: plans on different lanes overlap"""
        self.eng.check(self.eng.lib.gams_wave_plan_set_lane(self.eng.h, self.p, lane))

    def set_taper(self, mode):
        """-1 auto (on for depth 1), 0 off (passes in flight), 1 on: the launch ends in smaller tiles"""
        self.eng.check(self.eng.lib.gams_wave_plan_set_taper(self.eng.h, self.p, mode))

    def set_pipelined(self, on):
        """an event behind every run; peaks()/dense() then wait for that run only (plans in flight on lanes)"""
        self.eng.check(self.eng.lib.gams_wave_plan_set_pipelined(self.eng.h, self.p, int(on)))

    def peaks_count(self):
        """wait for the selected run, pack and fetch its peaks into the plan's page-locked buffer; -> how many
        (the records stay in plan-owned memory: what a C host reads in place, no copy into numpy)"""
        ptr, n = C.c_void_p(), C.c_uint64()
        self.eng.check(self.eng.lib.gams_wave_peaks(self.eng.h, self.p, C.byref(ptr), C.byref(n)))
        return int(n.value)

    def rows_setup(self, chr_names, chr_starts, coverage=0.2):
        """device-made TSV rows (gams_wave_rows_*): chromosome name / first coordinate per ctg, --coverage"""
        n = len(chr_names)
        self._row_names = (C.c_char_p * max(n, 1))(*[x.encode() for x in chr_names])
        self._row_starts = np.asarray(chr_starts, np.int32)
        self.eng.check(self.eng.lib.gams_wave_rows_setup(self.eng.h, self.p, self._row_names, self._row_starts.ctypes.data,
                                                         coverage))

    def rows_begin(self):
        self.eng.check(self.eng.lib.gams_wave_rows_begin(self.eng.h, self.p))

    def rows_end(self, copy=True):
        """-> (text bytes, ctg offsets); copy=False: only the byte count (the text stays in plan-owned memory)"""
        txt, n, off = C.c_void_p(), C.c_uint64(), C.c_void_p()
        self.eng.check(self.eng.lib.gams_wave_rows_end(self.eng.h, self.p, C.byref(txt), C.byref(n), C.byref(off)))
        if not copy:
            return int(n.value)
        n_ctg = len(self._row_names) if self._row_starts.size else 0
        offs = np.frombuffer((C.c_uint64 * (n_ctg + 1)).from_address(off.value), np.uint64).copy()
        return (C.string_at(txt.value, n.value) if n.value else b""), offs

    def signal_text(self, chr_names, chr_starts, copy=True):
        """`wave --signal` rows of the selected run as text from the device -> (text bytes, ctg offsets) (copy=False: byte count)"""
        n = len(chr_names)
        names = (C.c_char_p * max(n, 1))(*[x.encode() for x in chr_names])
        starts = np.asarray(chr_starts, np.int32)
        txt, nb, off = C.c_void_p(), C.c_uint64(), C.c_void_p()
        self.eng.check(self.eng.lib.gams_wave_signal_text(self.eng.h, self.p, names, starts.ctypes.data, C.byref(txt), C.byref(nb),
                                                          C.byref(off)))
        if not copy:
            return int(nb.value)
        offs = np.frombuffer((C.c_uint64 * (n + 1)).from_address(off.value), np.uint64).copy()
        return (C.string_at(txt.value, nb.value) if nb.value else b""), offs

    def set_taper_shape(self, pct4, pct8=None):
        """size of the tapered launch's two tails in % of a round of workgroup slots (default 25 / 50)"""
        if pct8 is None:                      # tools/ab_plans.py passes one integer: pct4 * 1000 + pct8
            pct4, pct8 = divmod(int(pct4), 1000)
        self.eng.check(self.eng.lib.gams_wave_plan_set_taper_shape(self.eng.h, self.p, pct4, pct8))

    def set_queue_threads(self, n):
        """host threads run_n queues a long batch of passes from (1..4)"""
        self.eng.check(self.eng.lib.gams_wave_plan_set_queue_threads(self.eng.h, self.p, n))

    def kernel_name(self):
        """the kernel that does this plan's work, as rocprofv3 --kernel-trace names it (gams_wave_plan_kernel_name)"""
        buf = C.create_string_buffer(160)
        self.eng.check(self.eng.lib.gams_wave_plan_kernel_name(self.eng.h, self.p, buf, len(buf)))
        return buf.value.decode()

    def select(self, age):
        """point peaks()/dense()/exact_count() at the run `age` runs before the most recent one"""
        self.eng.check(self.eng.lib.gams_wave_plan_select(self.eng.h, self.p, age))

    def run_n(self, n):
        self.eng.check(self.eng.lib.gams_wave_run_n(self.eng.h, self.p, n))

    def run(self):
        self.eng.check(self.eng.lib.gams_wave_run(self.eng.h, self.p))

    def peaks(self):
        ptr, n = C.c_void_p(), C.c_uint64()
        self.eng.check(self.eng.lib.gams_wave_peaks(self.eng.h, self.p, C.byref(ptr), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, PEAK_DTYPE)
        buf = (C.c_char * (n.value * PEAK_DTYPE.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=PEAK_DTYPE).copy()

    def dense(self, i):
        n = self.ctg_windows(i)
        cnt = np.zeros(max(n, 1), np.uint32)
        sig = np.zeros(max(n, 1), np.int8)
        self.eng.check(self.eng.lib.gams_wave_dense(self.eng.h, self.p, i, cnt.ctypes.data, sig.ctypes.data))
        return cnt[:n], sig[:n]

    def stamps(self):
        """Run once with phase stamps on; returns (mean cycles per phase[7], span cycles)."""
        self.eng.check(self.eng.lib.gams_wave_plan_set_stamps(self.eng.h, self.p, 1))
        self.run()
        m = (C.c_double * 8)()
        span = C.c_uint64()
        self.eng.check(self.eng.lib.gams_wave_stamps(self.eng.h, self.p, m, C.byref(span)))
        self.eng.check(self.eng.lib.gams_wave_plan_set_stamps(self.eng.h, self.p, 0))
        span_ticks, sum_ticks = span.value >> 32, (span.value & 0xffffffff) * 16
        info = dict(clock_ghz=m[7], span_us=span_ticks / 100.0,
                    mean_resident_wg=(sum_ticks / span_ticks) if span_ticks else 0.0)
        return list(m)[:7], info

    def set_guard(self, safety=1.5, all_exact=False):
        """diagnostics: safety factor of the guard band / every window through the exact path"""
        self.eng.check(self.eng.lib.gams_wave_plan_set_guard(self.eng.h, self.p, safety, int(all_exact)))

    def exact_count(self):
        n = C.c_uint64()
        self.eng.check(self.eng.lib.gams_wave_exact_count(self.eng.h, self.p, C.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "p", None):
            self.eng.lib.gams_wave_plan_destroy(self.eng.h, self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
