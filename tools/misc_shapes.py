#!/usr/bin/env python3
"""tools/misc_shapes.py -- ac_intg_dump and ac_mv_avg outside their bench rows: block lengths, channel counts, window lengths and modes, frame
lengths, sample / output widths.  512 objects x 2^20 samples per call; one line per shape: ms per call, TB/s of read + written bytes, path."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ac_dsp_amd as A  # noqa: E402

n_obj, n, K = 512, 1 << 20, 5
F = A.Fmt


_settled = [False]


def timed(fn):
    if not _settled[0]:          # the first row of a process met a cold shader clock (0.51 - 0.60 where later rows of the same shape run 0.68): ~0.3 s of it first
        t = time.perf_counter()
        while time.perf_counter() - t < 0.3:
            fn()
        torch.cuda.synchronize()
        _settled[0] = True
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K


which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("both", "intg"):
    for ns, chn, fin, fa, fo in ((64, 4, F(16, 8), F(32, 16), F(32, 16)), (8, 4, F(16, 8), F(32, 16), F(32, 16)), (1000, 4, F(16, 8), F(40, 24), F(40, 24)),
                                 (64, 1, F(16, 8), F(32, 16), F(32, 16)), (64, 7, F(16, 8), F(32, 16), F(32, 16)), (64, 16, F(16, 8), F(32, 16), F(16, 8, True, "RND", "SAT")),
                                 (64, 4, F(32, 16), F(48, 32), F(48, 32)), (64, 4, F(12, 4), F(24, 12, True, "TRN", "SAT"), F(24, 12)), (256, 2, F(16, 8), F(32, 16), F(32, 16)),
                                 (1024, 4, F(32, 16), F(64, 32), F(64, 32))):   # last: the types of the header's usage example (ac_intg_dump.h:47-51)
        try:
            eng = A.IntgDump(ns, chn, fin, fa, fo, n_objects=n_obj)
        except Exception as e:  # noqa: BLE001
            print("intg_dump NS=%d CHN=%d rejected: %s" % (ns, chn, str(e)[:60]))
            continue
        blocks = n // (ns * chn)
        nsv = np.full(blocks, ns, dtype=np.int64)
        x = torch.empty((n_obj, blocks * ns * chn), dtype=A.torch_dtype_for(fin), device="cuda")
        A.fill_stimulus(x, 1, min(fin.W, 16))
        out = eng.run(x, nsv)
        ms = timed(lambda: eng.run(x, nsv))
        gb = (x.numel() * x.element_size() + out.numel() * out.element_size()) / 1e9
        print("intg_dump NS=%4d CHN=%2d <%d,%d> -> ACC <%d,%d,%s,%s> -> <%d,%d>  %.3f ms  %.2f TB/s  %.3f of 8 TB/s" %
              (ns, chn, fin.W, fin.I, fa.W, fa.I, fa.Q, fa.O, fo.W, fo.I, ms, gb / ms, gb / ms / 8), flush=True)
        del eng, x, out
if which in ("both", "mvavg"):
    for taps, mode, ns, fin, fa, fo in ((9, "MIRROR", 1024, F(16, 8), F(40, 18), F(16, 8, True, "RND", "SAT")), (9, "WIN", 1024, F(16, 8), F(40, 18), F(16, 8, True, "RND", "SAT")),
                                        (9, "CLIP", 1024, F(16, 8), F(40, 18), F(16, 8, True, "RND", "SAT")), (3, "MIRROR", 1024, F(16, 8), F(40, 18), F(16, 8, True, "RND", "SAT")),
                                        (33, "MIRROR", 1024, F(16, 8), F(40, 18), F(16, 8, True, "RND", "SAT")), (65, "MIRROR", 1024, F(16, 8), F(40, 18), F(16, 8, True, "RND", "SAT")),
                                        (13, "WIN", 1024, F(16, 8), F(40, 18), F(16, 8, True, "RND", "SAT")), (29, "WIN", 1024, F(16, 8), F(40, 18), F(32, 12)),   # output frames off a 16-byte boundary
                                        (9, "MIRROR", 1004, F(16, 8), F(40, 18), F(16, 8, True, "RND", "SAT")),   # frames that are no multiple of 8 samples: the sliding-window kernel
                                        (9, "MIRROR", 128, F(16, 8), F(40, 18), F(16, 8, True, "RND", "SAT")), (9, "MIRROR", 4096, F(16, 8), F(40, 18), F(16, 8, True, "RND", "SAT")),
                                        (9, "MIRROR", 1000, F(16, 8), F(40, 18), F(16, 8, True, "RND", "SAT")), (9, "MIRROR", 1024, F(16, 8), F(40, 18), F(32, 12)),
                                        (9, "MIRROR", 1024, F(16, 8), F(40, 18), F(40, 18)), (9, "MIRROR", 1024, F(12, 4), F(30, 10), F(12, 4, True, "RND", "SAT")),
                                        (9, "MIRROR", 1024, F(32, 16), F(56, 30), F(32, 16, True, "RND", "SAT")), (9, "MIRROR", 1024, F(16, 8), F(24, 10), F(16, 8, True, "RND", "SAT")),
                                        (9, "MIRROR", 1024, F(32, 16), F(16, 2), F(64, 32))):   # last: the types of the header's usage example (ac_mv_avg.h:47-51; COEFF <32,16> below)
        try:
            fcf = F(32, 16) if (fa.W == 16 and fin.W == 32) else F(16, 2)
            eng = A.MvAvg(4096, taps, mode, fin, fcf, fa, fo, n_objects=n_obj)
        except Exception as e:  # noqa: BLE001
            print("mv_avg TAPS=%d %s rejected: %s" % (taps, mode, str(e)[:60]))
            continue
        w = np.hanning(taps + 2)[1:-1]
        eng.set_coeffs(np.round(w / w.sum() * 2.0 ** (fcf.W - fcf.I - (2 if fcf.W == 32 else 0))).astype(np.int64))
        frames = n // ns
        x = torch.empty((n_obj, frames * ns), dtype=A.torch_dtype_for(fin), device="cuda")
        A.fill_stimulus(x, 1, min(fin.W, 16))
        out = eng.run(x, ns)
        ms = timed(lambda: eng.run(x, ns))
        gb = (x.numel() * x.element_size() + out.numel() * out.element_size()) / 1e9
        print("mv_avg TAPS=%2d %-6s frames of %4d <%d,%d> -> ACC <%d,%d> -> <%d,%d,%s,%s>  %.3f ms  %.2f TB/s  %.3f of 8 TB/s  path %s" %
              (taps, mode, ns, fin.W, fin.I, fa.W, fa.I, fo.W, fo.I, fo.Q, fo.O, ms, gb / ms, gb / ms / 8, eng.path), flush=True)
        del eng, x, out
