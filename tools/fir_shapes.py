#!/usr/bin/env python3
"""tools/fir_shapes.py -- ac_fir_* outside the BASELINE rows: the reference testbench's own types, 32-bit samples, narrow types, every ftype,
the lossy and the saturating accumulator classes.  512 channels x 2^19 samples per call; one line per shape: ms per call (events on the
current stream), TB/s of read + written bytes, fraction of 8 TB/s, kernel path.  Looks for cliffs between the compiled shapes."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ac_dsp_amd as A  # noqa: E402
from bench import windowed_sinc_raw  # noqa: E402

n_ch, n, K = 512, 1 << 19, 5
F = A.Fmt
CASES = [
    # name, taps, ftype, IN, COEFF, ACC, OUT
    ("cfg1 types, 63 taps, OUT = ACC <38,10>", 63, "FOLD_ODD", F(16, 2), F(16, 2), F(38, 10), F(38, 10)),
    ("cfg1 types, 63 taps, OUT <16,2,RND,SAT>", 63, "FOLD_ODD", F(16, 2), F(16, 2), F(38, 10), F(16, 2, True, "RND", "SAT")),
    ("reference testbench types, 29 taps", 29, "FOLD_ODD", F(16, 8), F(32, 16), F(64, 32), F(64, 32)),
    ("rtest load types <32,16> x <32,16> -> <64,32>, 27 taps", 27, "FOLD_ODD", F(32, 16), F(32, 16), F(64, 32), F(64, 32)),
    ("rtest prog types <28,6> x <23,7> -> <64,32>, 27 taps", 27, "FOLD_ODD", F(28, 6), F(23, 7), F(64, 32), F(64, 32)),
    ("prog types, 27 taps SHIFT_REG (class B)", 27, "SHIFT_REG", F(28, 6), F(23, 7), F(64, 32), F(64, 32)),
    ("prog types, 27 taps FOLD_ODD, OUT <32,8,RND,SAT>", 27, "FOLD_ODD", F(28, 6), F(23, 7), F(64, 32), F(32, 8, True, "RND", "SAT")),
    ("prog types, 127 taps SHIFT_REG (class B)", 127, "SHIFT_REG", F(28, 6), F(23, 7), F(64, 32), F(64, 32)),
    ("<16,8> x <32,16> -> ACC <48,28> lossy s=4, 29 taps", 29, "FOLD_ODD", F(16, 8), F(32, 16), F(48, 28), F(48, 28)),
    ("<16,8> x <32,16>, 29 taps, OUT <32,16,RND,SAT>", 29, "SHIFT_REG", F(16, 8), F(32, 16), F(60, 36), F(32, 16, True, "RND", "SAT")),
    ("<16,8> x <32,16>, 127 taps, OUT <16,8,RND,SAT>", 127, "SHIFT_REG", F(16, 8), F(32, 16), F(60, 36), F(16, 8, True, "RND", "SAT")),
    ("DDC stage types, 127 taps <36,21> -> <32,17>", 127, "SHIFT_REG", F(36, 21), F(16, 1), F(59, 29), F(32, 17, True, "RND", "SAT")),
    ("<32,16> x <32,16>, 64 taps, ACC = OUT <64,32>", 64, "SHIFT_REG", F(32, 16), F(32, 16), F(64, 32), F(64, 32)),
    ("<32,16> x <16,2>, 127 taps, OUT <32,16,RND,SAT>", 127, "SHIFT_REG", F(32, 16), F(16, 2), F(56, 26), F(32, 16, True, "RND", "SAT")),
    ("<12,1> x <12,1>, 127 taps, OUT <12,1,RND,SAT>", 127, "SHIFT_REG", F(12, 1), F(12, 1), F(31, 8), F(12, 1, True, "RND", "SAT")),
    ("<8,1> x <8,1>, 63 taps, OUT <8,1,RND,SAT>", 63, "SHIFT_REG", F(8, 1), F(8, 1), F(22, 7), F(8, 1, True, "RND", "SAT")),
    ("<16,2> unsigned samples, 127 taps", 127, "SHIFT_REG", F(16, 2, False), F(16, 2), F(40, 12), F(16, 3, True, "RND", "SAT")),
    ("255 taps TRANSPOSED", 255, "TRANSPOSED", F(16, 2), F(16, 2), F(40, 12), F(16, 2, True, "RND", "SAT")),
    ("255 taps C_BUFF", 255, "C_BUFF", F(16, 2), F(16, 2), F(40, 12), F(16, 2, True, "RND", "SAT")),
    ("255 taps ROTATE_SHIFT", 255, "ROTATE_SHIFT", F(16, 2), F(16, 2), F(40, 12), F(16, 2, True, "RND", "SAT")),
    ("254 taps FOLD_EVEN", 254, "FOLD_EVEN", F(16, 2), F(16, 2), F(40, 12), F(16, 2, True, "RND", "SAT")),
    ("255 taps FOLD_ODD", 255, "FOLD_ODD", F(16, 2), F(16, 2), F(40, 12), F(16, 2, True, "RND", "SAT")),
    ("255 taps, OUT <24,6,RND,SAT> (4-byte containers)", 255, "SHIFT_REG", F(16, 2), F(16, 2), F(40, 12), F(24, 6, True, "RND", "SAT")),
    ("255 taps, OUT <32,12,TRN,WRAP> (4-byte containers)", 255, "SHIFT_REG", F(16, 2), F(16, 2), F(40, 12), F(32, 12)),
    ("255 taps, OUT = ACC <40,12> (8-byte containers)", 255, "SHIFT_REG", F(16, 2), F(16, 2), F(40, 12), F(40, 12)),
    ("255 taps, OUT <16,2,RND_ZERO,SAT>", 255, "SHIFT_REG", F(16, 2), F(16, 2), F(40, 12), F(16, 2, True, "RND_ZERO", "SAT")),
    ("255 taps, OUT <16,2,RND_CONV,SAT_SYM>", 255, "SHIFT_REG", F(16, 2), F(16, 2), F(40, 12), F(16, 2, True, "RND_CONV", "SAT_SYM")),
    ("63 taps, lossy ACC <24,8,TRN,WRAP> (class B)", 63, "SHIFT_REG", F(16, 2), F(16, 2), F(24, 8), F(16, 2, True, "RND", "SAT")),
    ("63 taps, lossy ACC <24,8,RND,WRAP> (class B)", 63, "SHIFT_REG", F(16, 2), F(16, 2), F(24, 8, True, "RND", "WRAP"), F(16, 2, True, "RND", "SAT")),
    ("63 taps, lossy ACC <32,12,TRN,WRAP>: 8 bits dropped (class B)", 63, "SHIFT_REG", F(16, 2), F(16, 2), F(32, 12), F(16, 2, True, "RND", "SAT")),
    ("127 taps, lossy ACC <32,8,RND,WRAP>: 4 bits dropped (class B)", 127, "SHIFT_REG", F(16, 2), F(16, 2), F(32, 8, True, "RND", "WRAP"), F(16, 2, True, "RND", "SAT")),
    ("63 taps, saturating ACC <30,4,TRN,SAT> (class C)", 63, "SHIFT_REG", F(16, 2), F(16, 2), F(30, 4, True, "TRN", "SAT"), F(16, 2, True, "RND", "SAT")),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, taps, ftype, fin, fc, fa, fo in CASES:
    if only and only not in name:
        continue
    try:
        eng = A.Fir(taps, ftype, fin, fc, fa, fo, n_channels=n_ch, kind="load")
    except Exception as e:  # noqa: BLE001
        print("%-52s rejected: %s" % (name, str(e)[:80]))
        continue
    c = windowed_sinc_raw(taps | 1, 0.1, fc.F)[:taps]
    lim = (1 << (fc.W - 1)) - 1
    eng.set_coeffs(np.clip(c, -lim, lim))
    x = torch.empty((n_ch, n), dtype=A.torch_dtype_for(fin), device="cuda")
    A.fill_stimulus(x, 0xACD5, fin.W - (0 if fin.S else 1))
    y = torch.empty((n_ch, n), dtype=A.torch_dtype_for(fo), device="cuda")
    cls = eng.path
    slow = cls in ("generic", "exact_order")
    reps = 1 if slow else K
    for _ in range(1 if slow else 3):
        eng.run(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.run(x, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = (x.numel() * x.element_size() + y.numel() * y.element_size()) / 1e9
    print("%-52s %8.3f ms  %5.2f TB/s  %.3f of 8 TB/s  path %s" % (name, ms, gb / ms, gb / ms / 8, eng.kernel), flush=True)
    del eng, x, y
