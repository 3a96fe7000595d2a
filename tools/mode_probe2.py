#!/usr/bin/env python3
"""tools/mode_probe2.py -- follow-up of mode_probe.py (the speed of a streaming row follows the PAIR of buffers): input and output
carved out of ONE allocation, output placed at a swept byte offset D behind the input.  Time against D shows which address bits
of the read-stream / write-stream distance matter."""
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
XB = NCH * N * 2
YROW = N // 8 + 8
YB = NCH * YROW * 2
pool = torch.empty(XB + YB + (1 << 30), dtype=torch.uint8, device=dev)
x = pool[:XB].view(torch.int16).view(NCH, N)
A.fill_stimulus(x, 0xACD5, 16, ch0=0)


def t_of(D, reps=8):
    y = pool[XB + D: XB + D + YB].view(torch.int16).view(NCH, YROW)
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


for _ in range(200):
    t_of(0, 1)
print("pool at %#x" % pool.data_ptr())
for name, step, cnt in (("2 MB", 2 << 20, 48), ("128 KB", 128 << 10, 32), ("4 KB", 4 << 10, 32), ("256 B", 256, 16)):
    ts = [t_of(k * step) for k in range(cnt)]
    print("step %s: " % name + " ".join("%.3f" % t for t in ts))
ts = [t_of(0) for _ in range(6)]
print("repeat D=0: " + " ".join("%.3f" % t for t in ts))
# same process, same input: outputs in separate allocations
for i in range(6):
    ysep = torch.empty((NCH, YROW), dtype=torch.int16, device=dev)
    spacer = torch.empty((5 + 13 * i) << 20, dtype=torch.uint8, device=dev)
    for _ in range(2):
        eng.run(x, ysep)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        eng.run(x, ysep)
    e1.record()
    torch.cuda.synchronize()
    print("separate output allocation %d at %#x: %.3f ms" % (i, ysep.data_ptr(), e0.elapsed_time(e1) / 8))
    keep = (ysep, spacer) if i % 2 else None
