#!/usr/bin/env python3
"""tools/mode_probe4.py [pool GB] [step MB] -- the output block of the poly_dec row placed at distances of up to many GB behind the
input block inside ONE allocation (mode_probe2.py: distances below 100 MB change nothing; separate allocations differ by GBs)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ac_dsp_amd as A
from bench import windowed_sinc_raw

dev = torch.device("cuda", 0)
NCH, N = 1024, 1 << 22
POOL = (int(sys.argv[1]) if len(sys.argv) > 1 else 16) << 30
STEP = (int(sys.argv[2]) if len(sys.argv) > 2 else 32) << 20
fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
eng = A.PolyDec(16, 8, fin, fc, fa, fo, n_channels=NCH, device=0)
hh = np.concatenate([windowed_sinc_raw(127, 0.05, fc.F), [0]])
eng.set_coeffs(np.array([hh[df + tp * 8] for df in range(8) for tp in range(16)], dtype=np.int64))
XB = NCH * N * 2
YROW = N // 8 + 8
YB = NCH * YROW * 2
pool = torch.empty(POOL, dtype=torch.uint8, device=dev)
x = pool[:XB].view(torch.int16).view(NCH, N)
A.fill_stimulus(x, 0xACD5, 16, ch0=0)


def t_of(D, reps=6):
    y = pool[XB + D: XB + D + YB].view(torch.int16).view(NCH, YROW)
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
print("pool at %#x, %d GB, step %d MB; one line per GB of distance" % (pool.data_ptr(), POOL >> 30, STEP >> 20))
D = 0
line = []
while XB + D + YB <= POOL:
    line.append(t_of(D))
    D += STEP
    if D % (1 << 30) == 0:
        print("%3d GB: " % ((D >> 30) - 1) + " ".join("%.2f" % t for t in line))
        line = []
if line:
    print("  tail: " + " ".join("%.2f" % t for t in line))
