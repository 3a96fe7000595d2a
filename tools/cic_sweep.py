#!/usr/bin/env python3
"""tools/cic_sweep.py -- ac_cic_dec_full over parameter sets other than the BASELINE ones (which kernel they get, and how far from the roofline):
4096 ch x 2^20 samples, ms per launch and fraction of 8 TB/s on the algorithmic bytes (in + out / R)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ac_dsp_amd as A

dev = torch.device("cuda", 0)
NCH, N = 4096, 1 << 20
SMALL = ((32, 16, 8, 1, 5), (32, 16, 7, 2, 4), (32, 16, 4, 1, 3), (32, 16, 16, 1, 5), (32, 16, 32, 1, 4), (32, 16, 10, 1, 5), (16, 1, 16, 1, 5),
         (16, 1, 8, 1, 4), (16, 1, 64, 1, 3), (16, 1, 5, 1, 6), (24, 8, 8, 2, 3),
         (16, 1, 3, 1, 5), (32, 16, 3, 1, 5), (16, 1, 6, 1, 5), (32, 16, 6, 1, 5), (16, 1, 10, 1, 4), (16, 1, 12, 1, 5), (16, 1, 20, 1, 5), (32, 16, 5, 1, 5), (16, 1, 7, 2, 4))
# `large`: the rates a CIC is deployed at (36 / 42 / 44 / 49 / 252: the second set of stage-1 rates; 37: a prime, recurrence kernel), every (R, N) the reference's int power<> admits ((R M)^N < 2^31), 16- and 32-bit samples
LARGE = tuple((W, I, R, M, Ns) for (W, I) in ((16, 1), (32, 16)) for R in (32, 36, 42, 44, 49, 64, 100, 128, 250, 252, 255, 37) for (M, Ns) in ((1, 3), (1, 4), (1, 5), (2, 3))
              if (R * M) ** Ns < 2 ** 31)
QUICK = ((16, 1, 32, 1, 3), (16, 1, 32, 1, 5), (16, 1, 64, 1, 4), (16, 1, 128, 2, 3), (32, 16, 32, 1, 5), (32, 16, 64, 1, 3), (32, 16, 128, 1, 4))
SETS = LARGE if "large" in sys.argv[1:] else (QUICK if "quick" in sys.argv[1:] else SMALL)
for (W, I, R, M, Ns) in SETS:
    fin = A.Fmt(W, I)
    it = A.Cic(False, R, M, Ns, fin, fin, n_channels=1, device=0).int_type
    fo = A.Fmt(it.W, it.I)
    eng = A.Cic(False, R, M, Ns, fin, fo, n_channels=NCH, device=0)
    dt = A.torch_dtype_for(fin)
    x = torch.empty((NCH, N), dtype=dt, device=dev)
    A.fill_stimulus(x, 0xACD5, W if W <= 32 else 32, ch0=0)
    y = torch.empty((NCH, (N // R + 8 + 7) // 8 * 8), dtype=A.torch_dtype_for(fo), device=dev)
    for _ in range(10):
        eng.run(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        eng.run(x, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    nbytes = NCH * N * (x.element_size() + y.element_size() / R)
    print("IN <%d,%d> R %2d M %d N %d -> INT %2d bits  path %-12s %7.3f ms  %.3f of 8 TB/s" % (W, I, R, M, Ns, it.W, getattr(eng, "path", "?"), ms, nbytes / (ms * 1e-3) / 8e12))
    del eng, x, y
