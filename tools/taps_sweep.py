#!/usr/bin/env python3
"""tools/taps_sweep.py [dense] -- the int8 MFMA FIR over the tap count (1024 ch x 2^20 samples, <16,2> types, OUT <16,2,RND,SAT>): ms per
launch, the kernel path the plan chose and the MFMAs issued per 1024 outputs; looks for cliffs between the compiled shapes."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ac_dsp_amd as A
from bench import windowed_sinc_raw

dense = len(sys.argv) > 1 and sys.argv[1] == "dense"
dev = torch.device("cuda", 0)
NCH, N = 1024, 1 << 20
fin, fc, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(16, 2, True, "RND", "SAT")
x = torch.empty((NCH, N), dtype=torch.int16, device=dev)
A.fill_stimulus(x, 0xACD5, 16, ch0=0)
y = torch.empty((NCH, N), dtype=torch.int16, device=dev)
TAPS = [int(t) for t in os.environ["TAPS"].split(",")] if os.environ.get("TAPS") else (31, 63, 127, 191, 255, 287, 289, 319, 383, 447, 511, 639, 767, 895, 991, 993, 1023)
for taps in TAPS:
    fa = A.Fmt(42, 14)
    eng = A.Fir(taps, "SHIFT_REG", fin, fc, fa, fo, n_channels=NCH, kind="load", device=0)
    if dense:
        c = np.random.default_rng(1).integers(-32768, 32640, size=taps, dtype=np.int64)
    else:
        c = windowed_sinc_raw(taps, 0.1 * 255 / max(taps, 255), fc.F)
    eng.set_coeffs(c)
    for _ in range(30):
        eng.run(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        eng.run(x, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    issued = eng.mfma_issued() if hasattr(eng, "mfma_issued") else -1
    print("taps %4d  %-10s %6.3f ms  %5.2f ns/sample/ktap  issued MFMAs/step %s" % (taps, eng.path, ms, ms * 1e6 / (NCH * N) / taps * 1000, issued))
    del eng
