#!/usr/bin/env python3
"""tools/intr_shapes.py -- the interpolating MFMA kernel (fir_up_kernel) outside the two bench rows: ac_cic_intr_full at R = 4 / 8 / 16 on
16- and 32-bit inputs, 1024 channels.  One line per shape: ms per call (events on the current stream) and TB/s of read + written bytes.
A/B two builds by running it once per build (ACDSP_LIB=...), alternating."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import ac_dsp_amd as A  # noqa: E402

n_ch = 1024
K = 10
M_OF = {7: 2}
for w_in, R, n in ((32, 4, 1 << 19), (32, 8, 1 << 18), (32, 16, 1 << 17), (16, 4, 1 << 19), (16, 8, 1 << 18), (16, 16, 1 << 17), (32, 7, 1 << 18), (16, 7, 1 << 18), (32, 3, 1 << 19), (32, 5, 1 << 18), (16, 6, 1 << 18)):
    fin = A.Fmt(w_in, w_in // 2)
    N = 5 if w_in == 32 or R < 16 else 4
    M = M_OF.get(R, 1)
    probe = A.Cic(True, R, M, N, fin, fin, n_channels=1)
    it = probe.int_type
    fout = A.Fmt(it.W, it.I)
    eng = A.Cic(True, R, M, N, fin, fout, n_channels=n_ch)
    x = torch.empty((n_ch, n), dtype=A.torch_dtype_for(fin), device="cuda")
    A.fill_stimulus(x, 1, w_in)
    y = torch.empty((n_ch, n * R + 64), dtype=A.torch_dtype_for(fout), device="cuda")
    for _ in range(3):
        eng.reset()
        out = eng.run(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        out = eng.run(x, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    gb = (x.numel() * x.element_size() + out.numel() * out.element_size()) / 1e9
    print("cic_intr N%d M%d R%-2d <%d> -> <%d,%d> (%d-byte outputs): %.3f ms  %.2f TB/s  path %s" % (N, M, R, w_in, it.W, it.I, out.element_size(), ms, gb / ms, eng.path))
    del eng, x, y, out
