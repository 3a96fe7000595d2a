#!/usr/bin/env python3
"""tools/mode_probe.py -- where do the two speeds of the streaming rows come from (poly_dec 1.69 / 1.83 ms, fused DDC 1.80 / 1.95 ms,
config 3 14.4 / 15.3 ms: every box shows both, a process stays in one)?  One process, one engine handle, several independently
allocated input / output buffer sets visited round-robin: if the speed follows the buffer set it is the physical placement of the
buffers; if it follows the process it is device state."""
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ac_dsp_amd as A
from bench import windowed_sinc_raw

dev = torch.device("cuda", 0)
WL = sys.argv[2] if len(sys.argv) > 2 else "polydec"
fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
if WL == "fir255":      # the headline row: 255 taps, 1024 channels x 2^20 samples, int16 -> int16
    NCH, N, NOUT = 1024, 1 << 20, 1 << 20
    eng = A.Fir(255, "SHIFT_REG", fin, fc, fa, fo, n_channels=NCH, kind="load", device=0)
    eng.set_coeffs(windowed_sinc_raw(255, 0.1, fc.F))
else:                   # poly_dec: 128 taps, decimate by 8, 1024 channels x 2^22 samples
    NCH, N = 1024, 1 << 22
    NOUT = N // 8 + 8
    eng = A.PolyDec(16, 8, fin, fc, fa, fo, n_channels=NCH, device=0)
    hh = np.concatenate([windowed_sinc_raw(127, 0.05, fc.F), [0]])
    eng.set_coeffs(np.array([hh[df + tp * 8] for df in range(8) for tp in range(16)], dtype=np.int64))
sets = []
keep = []
nsets = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for i in range(nsets):
    x = torch.empty((NCH, N), dtype=torch.int16, device=dev)
    A.fill_stimulus(x, 0xACD5 + i, 16, ch0=0)
    y = torch.empty((NCH, NOUT), dtype=torch.int16, device=dev)
    keep.append(torch.empty((37 + 11 * i) << 20, dtype=torch.uint8, device=dev))   # odd-sized spacer: moves the next set
    sets.append((x, y))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    eng.run(*sets[0])
    torch.cuda.synchronize()
for rnd in range(4):
    out = []
    for i, (x, y) in enumerate(sets):
        for _ in range(3):
            eng.run(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.run(x, y)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10)
    print("round %d: " % rnd + "  ".join("set%d %.3f ms (x %#x y %#x)" % (i, t, sets[i][0].data_ptr(), sets[i][1].data_ptr()) if rnd == 0 else "set%d %.3f" % (i, t)
                                          for i, t in enumerate(out)))
# (x_i, y_j) matrix: is it one buffer's placement or the pair's?
print("matrix rows x_i, cols y_j (ms):")
for i in range(nsets):
    row = []
    for j in range(nsets):
        x, y = sets[i][0], sets[j][1]
        for _ in range(2):
            eng.run(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            eng.run(x, y)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 8)
    print("  x%d: " % i + " ".join("%.3f" % t for t in row))
