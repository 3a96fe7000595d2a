#!/usr/bin/env python3
"""tools/mode_probe5.py <polydec|ddc|cic_dec|mvavg> [candidates] -- one input block, several separately allocated output blocks, the row's
time with each (the placement levels of profiles/r3_placement_modes.txt).  With the same allocation sequence two builds of the
library (ACDSP_LIB=...) see comparable placements in consecutive processes: an A/B per placement."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ac_dsp_amd as A
from bench import windowed_sinc_raw

dev = torch.device("cuda", 0)
WL = sys.argv[1] if len(sys.argv) > 1 else "polydec"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if WL == "polydec":
    NCH, N = 1024, 1 << 22
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
    eng = A.PolyDec(16, 8, fin, fc, fa, fo, n_channels=NCH, device=0)
    hh = np.concatenate([windowed_sinc_raw(127, 0.05, fc.F), [0]])
    eng.set_coeffs(np.array([hh[df + tp * 8] for df in range(8) for tp in range(16)], dtype=np.int64))
    xdt, xbits, ydt, NOUT = torch.int16, 16, torch.int16, N // 8 + 8
elif WL == "ddc":
    NCH, N = 4096, 1 << 20
    fc, fa, fo = A.Fmt(16, 1), A.Fmt(60, 30), A.Fmt(24, 9, True, "RND", "SAT")
    eng = A.Ddc(16, 1, 5, A.Fmt(16, 1), 127, "SHIFT_REG", fc, fa, fo, n_channels=NCH, kind="const", device=0)
    eng.set_coeffs(windowed_sinc_raw(127, 0.2, fc.F))
    xdt, xbits, ydt, NOUT = torch.int16, 16, torch.int32, N // 16 + 8
elif WL == "mvavg":
    NCH, N = 1024, 1 << 20
    fc = A.Fmt(16, 2)
    eng = A.MvAvg(1024, 9, "MIRROR", A.Fmt(16, 8), fc, A.Fmt(40, 18), A.Fmt(16, 8, True, "RND", "SAT"), n_objects=NCH, device=0)
    eng.set_coeffs(np.round(np.hanning(11)[1:-1] / np.hanning(11).sum() * 2.0 ** fc.F).astype(np.int64))
    xdt, xbits, ydt, NOUT = torch.int16, 16, torch.int16, N
else:
    NCH, N = 4096, 1 << 22
    eng = A.Cic(False, 8, 1, 5, A.Fmt(32, 16), A.Fmt(47, 31), n_channels=NCH, device=0)
    xdt, xbits, ydt, NOUT = torch.int32, 32, torch.int64, N // 8 + 8
x = torch.empty((NCH, N), dtype=xdt, device=dev)
A.fill_stimulus(x, 0xACD5, xbits, ch0=0)
ys, keep = [], []
for i in range(K):
    ys.append(torch.empty((NCH, NOUT), dtype=ydt, device=dev))
    keep.append(torch.empty((5 + 13 * i) << 20, dtype=torch.uint8, device=dev))


def run(y):
    if WL == "mvavg":
        eng.run(x, 1024, out=y)
    else:
        eng.run(x, y)


def t_of(y, reps):
    for _ in range(2):
        run(y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for _ in range(30):
    run(ys[0])
ts = [t_of(y, 6) for y in ys]
print("%s candidates: " % WL + " ".join("%.3f" % t for t in ts) + "   mean %.3f" % (sum(ts) / len(ts)))
