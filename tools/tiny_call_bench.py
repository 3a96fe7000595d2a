#!/usr/bin/env python3
"""tools/tiny_call_bench.py -- cost of one small host-side run() call (the drop-in headers' path: one channel, host buffers).
ac_fir_prog_coeffs::run is ONE sample per call (reference include/ac_dsp/ac_fir_prog_coeffs.h:281), so this is launch latency, not
throughput.  ACDSP_NO_PINNED=1 selects the former path (device staging buffers, two hipMemcpy2D, two launches, timing events)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import ac_dsp_amd as A  # noqa: E402

tag = "staging (ACDSP_NO_PINNED=1)" if os.environ.get("ACDSP_NO_PINNED") else "pinned zero-copy, fused state update"
for name, (fin, fc, fa, fo, ft) in {
        "prog testbench types <28,6> x <23,7> -> <64,32>, 27 taps, FOLD_ODD (exact-order kernel)":
            (A.Fmt(28, 6), A.Fmt(23, 7), A.Fmt(64, 32), A.Fmt(64, 32), "FOLD_ODD"),
        "<16,2> 255 taps SHIFT_REG -> <16,2,RND,SAT> (matrix-core kernel)":
            (A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT"), "SHIFT_REG")}.items():
    n_taps = 27 if fin.W == 28 else 255
    f = A.Fir(n_taps, ft, fin, fc, fa, fo, n_channels=1, kind="prog")
    f.set_coeffs(np.arange(n_taps, dtype=np.int64))
    for n in (1, 16, 256, 4096):
        x = np.zeros((1, n), dtype=np.int32 if fin.W > 16 else np.int16)
        for _ in range(20):
            f.run_host(x)
        t0 = time.perf_counter()
        for _ in range(400):
            f.run_host(x)
        dt = (time.perf_counter() - t0) / 400
        print("%-34s %s: 1 channel x %4d samples  %6.1f us per call" % (tag, name[:40], n, dt * 1e6))
