#!/usr/bin/env python3
"""tools/mode_probe3.py -- counters of the slow and the fast placement of one streaming row (poly_dec), for rocprofv3 --pmc:
one input block, eight candidate output allocations; the slowest and the fastest pair are then launched 8 times each, slow first
(the LAST 16 dispatches of fir_gen_fast_kernel in the counter CSV)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ac_dsp_amd as A
from bench import windowed_sinc_raw

dev = torch.device("cuda", 0)
NCH, N = 1024, 1 << 22
fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
eng = A.PolyDec(16, 8, fin, fc, fa, fo, n_channels=NCH, device=0)
hh = np.concatenate([windowed_sinc_raw(127, 0.05, fc.F), [0]])
eng.set_coeffs(np.array([hh[df + tp * 8] for df in range(8) for tp in range(16)], dtype=np.int64))
x = torch.empty((NCH, N), dtype=torch.int16, device=dev)
A.fill_stimulus(x, 0xACD5, 16, ch0=0)
ys, keep = [], []
for i in range(8):
    ys.append(torch.empty((NCH, N // 8 + 8), dtype=torch.int16, device=dev))
    keep.append(torch.empty((5 + 13 * i) << 20, dtype=torch.uint8, device=dev))


def t_of(y, reps):
    for _ in range(2):
        eng.run(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.run(x, y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for _ in range(100):
    eng.run(x, ys[0])
ts = [t_of(y, 6) for y in ys]
print("candidates: " + " ".join("%.3f" % t for t in ts))
slow, fast = int(np.argmax(ts)), int(np.argmin(ts))
print("slow = %d (%#x), fast = %d (%#x), x at %#x" % (slow, ys[slow].data_ptr(), fast, ys[fast].data_ptr(), x.data_ptr()))
for y in (ys[slow], ys[fast]):
    torch.cuda.synchronize()
    for _ in range(8):
        eng.run(x, y)
    torch.cuda.synchronize()
print("final: slow %.3f fast %.3f" % (t_of(ys[slow], 6), t_of(ys[fast], 6)))
