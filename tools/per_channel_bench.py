#!/usr/bin/env python3
"""tools/per_channel_bench.py -- a coefficient set per channel (a bank of ac_fir_prog_coeffs objects, each with its own low-pass) against one
shared set: 1024 ch x 2^20 samples, <16,2> types, OUT <16,2,RND,SAT>; ms per launch."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import ac_dsp_amd as A
from helpers import windowed_sinc

dev = torch.device("cuda", 0)
NCH, N = 1024, 1 << 20
fin, fc, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(16, 2, True, "RND", "SAT")
x = torch.empty((NCH, N), dtype=torch.int16, device=dev)
A.fill_stimulus(x, 0xACD5, 16, ch0=0)
y = torch.empty((NCH, N), dtype=torch.int16, device=dev)
for taps in (255, 511, 1023):
    for per in (False, True):
        eng = A.Fir(taps, "SHIFT_REG", fin, fc, A.Fmt(44, 16), fo, n_channels=NCH, kind="prog", coeffs_per_channel=per, device=0)
        if per:
            c = np.stack([windowed_sinc(taps, 0.02 + 0.0001 * ch, fc, gain=0.9) for ch in range(NCH)])
        else:
            c = windowed_sinc(taps, 0.05, fc, gain=0.9)
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
        print("taps %4d  %-22s path %-10s %6.3f ms  MFMAs/step %d" % (taps, "one set per channel" if per else "shared set", eng.path, e0.elapsed_time(e1) / 10, eng.mfma_issued()))
        del eng
