"""CPU-only checks of the drop-in boundary: libgams_gpu.so loads and exports exactly the
entry points include/gams_gpu.h declares; no compute call is made (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from gams_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("gams_gpu.h", "gams_gpu_diag.h")     # the binding surface, and the measurement / tuning entries


def declared_functions(headers=HEADERS):
    names = []
    for hdr in headers:
        src = open(os.path.join(ROOT, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(gams_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(_lib.PROTOTYPES)


def test_binding_surface_carries_no_lab_bench():
    """stamps, guard band, taper, tile size, kernel names and stopwatches are declared in
    gams_gpu_diag.h only; and the library leaves the process environment alone."""
    main = set(declared_functions(("gams_gpu.h",)))
    diag = set(declared_functions(("gams_gpu_diag.h",)))
    assert not (main & diag)
    for name in ("gams_wave_plan_set_stamps", "gams_wave_stamps", "gams_wave_stamps_raw", "gams_wave_plan_set_guard",
                 "gams_wave_plan_set_taper", "gams_wave_plan_set_tile", "gams_wave_plan_kernel_name",
                 "gams_gpu_timer_start", "gams_gpu_timer_stop", "gams_gpu_last_kernel_ms", "gams_wave_exact_count"):
        assert name in diag and name not in main, name
    for src in ("api.hip", "wave.hip", "sw.hip", "interval.hip", "gen.hip", "wave_kernels.hpp", "common.hpp"):
        text = open(os.path.join(ROOT, "gams_amd", "csrc", src)).read()
        assert "getenv(" not in text and "setenv(" not in text and "putenv(" not in text, src


def test_library_loads_and_exports_every_symbol():
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_window_count_matches_reference_loop():
    # window.rs:78-94
    lib = _lib.load()
    for length, size, step in [(230218, 100, 10), (99, 100, 10), (100, 100, 10), (109, 100, 10), (110, 100, 10),
                               (1000, 7, 3), (5, 1, 1)]:
        n, start = 0, 1
        while start + size - 1 <= length:
            n += 1
            start += step
        assert lib.gams_window_count(length, size, step) == n
    assert lib.gams_window_count(100, 0, 1) == -1 and lib.gams_window_count(100, 10, 0) == -1


def test_no_gpu_means_loud_failure():
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.gams_gpu_create(0, C.byref(h))
    if rc == 0:                                   # (the file is meant for the CPU container)
        lib.gams_gpu_destroy(h)
        pytest.skip("a GPU is present")
    assert rc == _lib.ENODEV
    from gams_amd import engine

    with pytest.raises(_lib.GamsError):
        engine.Engine(0)


def test_header_is_plain_c(tmp_path):
    """include/gams_gpu.h and gams_gpu_diag.h must compile as C99 (a Rust/C/Go host binds it as a C ABI) and link against the library."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text(
        '#include "gams_gpu.h"\n'
        '#include "gams_gpu_diag.h"\n'
        "int main(void) {\n"
        "    gams_wave_params_t p = {100, 10, 100u, 3.0f, 1.0f};\n"
        "    gams_peak_t pk = {0u, 0u, 0u, 0};\n"
        "    gams_gpu_t *h = 0;\n"
        "    (void)p; (void)pk;\n"
        "    /* no device here: the call must fail cleanly, not crash */\n"
        "    return gams_window_count(230218, 100, 10) == 23012 && gams_gpu_last_error(h) != 0 ? 0 : 1;\n"
        "}\n")
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           str(src), "-o", str(exe), "-L", os.path.join(root, "gams_amd"), "-lgams_gpu",
                           "-Wl,-rpath," + os.path.join(root, "gams_amd")])
    assert subprocess.call([str(exe)]) == 0
