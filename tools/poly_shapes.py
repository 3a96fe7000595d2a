#!/usr/bin/env python3
"""tools/poly_shapes.py -- ac_poly_dec / ac_poly_intr outside the two bench rows: decimation / interpolation factors 2 .. 16, 8 and 16 taps
per branch, 2- and 8-byte outputs, 1024 channels of ac_fixed<16,2> samples.  One line per shape: ms per call (events on the current stream),
TB/s of read + written bytes, kernel path.  A/B two builds by running it once per build (ACDSP_LIB=...), alternating."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ac_dsp_amd as A  # noqa: E402
import bench  # noqa: E402

n_ch, K = 1024, 10
fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12)
outs = (("<16,2,RND,SAT>", A.Fmt(16, 2, True, "RND", "SAT")), ("<40,12>", A.Fmt(40, 12)))


def timed(fn):
    for _ in range(3):
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
if which in ("both", "dec"):
    for df, tp in ((2, 16), (4, 16), (8, 16), (16, 16), (4, 8), (8, 8), (16, 8)):
        for oname, fo in outs:
            n = 1 << 22
            eng = A.PolyDec(tp, df, fin, fc, fa, fo, n_channels=n_ch)
            hh = np.concatenate([bench.windowed_sinc_raw(tp * df - 1, 0.4 / df, fc.F), [0]])
            eng.set_coeffs(np.array([hh[d + t * df] for d in range(df) for t in range(tp)], dtype=np.int64))
            x = torch.empty((n_ch, n), dtype=torch.int16, device="cuda")
            A.fill_stimulus(x, 1, 16)
            y = torch.empty((n_ch, n // df + 8), dtype=A.torch_dtype_for(fo), device="cuda")
            ms = timed(lambda: eng.run(x, y))
            gb = (x.numel() * 2 + n_ch * (n // df) * y.element_size()) / 1e9
            print("poly_dec  NTAPS=%2d DF=%2d -> %-15s %.3f ms  %.2f TB/s  %.3f of 8 TB/s  path %s" % (tp, df, oname, ms, gb / ms, gb / ms / 8, eng.path), flush=True)
            del eng, x, y
if which in ("both", "intr"):
    for ifac, tp in ((2, 16), (4, 16), (8, 16), (16, 16), (4, 8), (8, 8), (16, 8)):
        for oname, fo in outs:
            n = (1 << 21) // ifac
            csz = tp * ifac // 2
            eng = A.PolyIntr(tp, csz, ifac, "FOLD_EVEN", fin, fc, fa, fo, n_channels=n_ch)
            eng.set_ctrl(bench.windowed_sinc_raw(tp * ifac - 1, 0.4 / ifac, fc.F)[:csz], [1] * ifac, list(range(ifac)))
            x = torch.empty((n_ch, n), dtype=torch.int16, device="cuda")
            A.fill_stimulus(x, 1, 16)
            eng.run(x[:, :16])
            ms = timed(lambda: eng.run(x))
            gb = (x.numel() * 2 + n_ch * n * ifac * (A.torch_dtype_for(fo).itemsize)) / 1e9
            print("poly_intr NTAPS=%2d IF=%2d -> %-15s %.3f ms  %.2f TB/s  %.3f of 8 TB/s  path %s" % (tp, ifac, oname, ms, gb / ms, gb / ms / 8, eng.path), flush=True)
            del eng, x

if which in ("both", "intr", "hdr"):
    # the header's usage-example types (ac_poly_intr.h:44-48): <32,16> samples and coefficients, ACC = OUT = <64,32>
    f32, a64 = A.Fmt(32, 16), A.Fmt(64, 32)
    for ifac, tp in ((2, 16), (4, 16), (8, 16), (16, 16), (7, 16)):
        n = (1 << 20) // ifac
        csz = tp * ifac // 2
        eng = A.PolyIntr(tp, csz, ifac, "FOLD_EVEN", f32, f32, a64, a64, n_channels=n_ch)
        hh = bench.windowed_sinc_raw(tp * ifac - 1, 0.4 / ifac, 16)
        eng.set_ctrl(np.concatenate([hh, [0] * csz])[:csz], [1] * ifac, list(range(ifac)))
        x = torch.empty((n_ch, n), dtype=torch.int32, device="cuda")
        A.fill_stimulus(x, 1, 32)
        eng.run(x[:, :16])
        ms = timed(lambda: eng.run(x))
        gb = (x.numel() * 4 + n_ch * n * ifac * 8) / 1e9
        print("poly_intr <32,16> x <32,16> -> <64,32> NTAPS=%2d IF=%2d  %.3f ms  %.2f TB/s  %.3f of 8 TB/s  path %s" % (tp, ifac, ms, gb / ms, gb / ms / 8, eng.path), flush=True)
        del eng, x
