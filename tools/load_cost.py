#!/usr/bin/env python3
"""tools/load_cost.py -- what a fresh process pays before its first filtered sample: dlopen of libacdsp.so (one code object with every
kernel of every family), the first handle (HIP runtime + code object load on the device), the first launch (kernel upload / first-use
work) and the second launch, in ms.  Pure ctypes (no torch, no numpy): nothing else has touched the GPU when the clock starts.
bench.py runs it as a child process and prints the result as cold_start.load_ms."""
import ctypes as C
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ACDSP_NO_TORCH_INIT"] = "1"   # the package's import-time torch initialisation is not part of what a C caller pays
t0 = time.perf_counter()
spec = importlib.util.spec_from_file_location("acdsp_lib", os.path.join(ROOT, "ac_dsp_amd", "_lib.py"))
L = importlib.util.module_from_spec(spec)
spec.loader.exec_module(L)          # CDLL(libacdsp.so) happens here
t1 = time.perf_counter()
lib = L.lib


def chk(rc):
    if rc:
        raise SystemExit("acdsp error %d" % rc)


n_taps, n_ch, n = 255, 64, 1 << 14
F = L.Fmt
d = L.FirDesc(L.KINDS["load"], L.FTYPES["SHIFT_REG"], n_taps, n_ch, 0, F(16, 2), F(16, 2), F(40, 12), F(16, 2, True, "RND", "SAT"), 0, 0)
h = C.c_void_p()
chk(lib.acdsp_fir_create(C.byref(d), C.byref(h)))
t2 = time.perf_counter()
coeffs = (C.c_int64 * n_taps)(*[((i * 37) % 101) - 50 for i in range(n_taps)])
chk(lib.acdsp_fir_set_coeffs(h, coeffs))
x, y = C.c_void_p(), C.c_void_p()
chk(lib.acdsp_dev_alloc(0, n_ch * n * 2, C.byref(x)))
chk(lib.acdsp_dev_alloc(0, n_ch * n * 2, C.byref(y)))
chk(lib.acdsp_fill_stimulus(0, x, 2, n_ch, n, n, 1, 16, 0, 0, None))
chk(lib.acdsp_sync(0, None))
t3 = time.perf_counter()
chk(lib.acdsp_fir_run(h, x, n, n, y, n, None))
chk(lib.acdsp_sync(0, None))
t4 = time.perf_counter()
chk(lib.acdsp_fir_run(h, x, n, n, y, n, None))
chk(lib.acdsp_sync(0, None))
t5 = time.perf_counter()
print(json.dumps({"dlopen_ms": (t1 - t0) * 1e3, "first_handle_ms": (t2 - t1) * 1e3, "coeffs_alloc_stimulus_ms": (t3 - t2) * 1e3,
                  "first_run_ms": (t4 - t3) * 1e3, "second_run_ms": (t5 - t4) * 1e3, "library_bytes": os.path.getsize(L.LIB_PATH),
                  "total_to_first_output_ms": (t4 - t0) * 1e3}))
