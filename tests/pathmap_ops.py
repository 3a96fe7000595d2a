"""The descriptor grids of the operator families beside the FIR classes (tests/pathmap_grid.py): ac_cic_dec_full / ac_cic_intr_full (rates,
stage counts, differential delays, sample / output widths), ac_poly_dec (factors, widths), ac_mv_avg (window lengths, modes, widths) and
ac_intg_dump (channel counts, block lengths, widths).  These families decide their kernel per CALL (row alignment, call length), so a
descriptor resolves by one small aligned call and the path the handle reports afterwards."""
import numpy as np
import torch

import ac_dsp_amd as A

F = A.Fmt


def _buf(fmt, rows, n):
    """rows x n samples, 16-byte aligned rows (stride rounded up to 64 samples), small deterministic values"""
    stride = (n + 63) // 64 * 64
    x = torch.zeros((rows, stride), dtype=A.torch_dtype_for(fmt), device="cuda")
    A.fill_stimulus(x, 7, min(fmt.W, 12))
    return x[:, :n]


CIC_IN = {"s16": F(16, 1), "s32": F(32, 16), "u12": F(12, 4, False), "s24": F(24, 8), "s40": F(40, 20)}
CIC_OUT = ("int_type", "o16_rnd_sat", "o80")
CIC_R = (2, 3, 4, 5, 7, 8, 10, 12, 16, 20, 24, 32, 37, 48, 64, 100, 128, 250, 255, 256)
CIC_MN = ((1, 3), (2, 4), (1, 5))


def cic_grid():
    for interp in (0, 1):
        for iname, fin in CIC_IN.items():
            for R in CIC_R:
                for M, N in CIC_MN:
                    if (R * M) ** N >= 2 ** 31 or (interp and R > 64 and iname != "s16"):
                        continue
                    for oname in CIC_OUT:
                        if oname != "int_type" and (R not in (8, 32, 64) or (M, N) != (1, 3)):
                            continue
                        yield "cic|%s|%s|R%d|M%d|N%d|%s" % ("intr" if interp else "dec", iname, R, M, N, oname), (interp, fin, R, M, N, oname)


def cic_resolve(args):
    interp, fin, R, M, N, oname = args
    try:
        it = A.Cic(bool(interp), R, M, N, fin, fin).int_type
        fo = F(it.W, it.I) if oname == "int_type" else (F(16, 2, True, "RND", "SAT") if oname == "o16_rnd_sat" else F(80, 40))
        cic = A.Cic(bool(interp), R, M, N, fin, fo, n_channels=2)
    except A.AcdspError:
        return "rejected"
    n = 2048 if interp else 4 * 256 * 20 + 64 * R          # decimators: a few steps of the widest compiled stage-1 rate, whole periods
    n -= n % R if not interp else 0
    n = (n + 15) // 16 * 16
    cic.run(_buf(fin, 2, n))
    return cic.path


PD_TYPES = {"i16": (F(16, 2), F(16, 2), F(40, 12)), "i32": (F(32, 16), F(16, 2), F(56, 26)), "i16_lossy": (F(16, 2), F(16, 2), F(24, 8))}
PD_OUT = {"o16": F(16, 2, True, "RND", "SAT"), "o_acc": None}


def polydec_grid():
    for tname, (fin, fc, fa) in PD_TYPES.items():
        for df in (2, 3, 4, 8, 16):
            for tp in (8, 16):
                for oname, fo in PD_OUT.items():
                    yield "polydec|%s|DF%d|T%d|%s" % (tname, df, tp, oname), (fin, fc, fa, fo or F(fa.W, fa.I), df, tp)


def polydec_resolve(args):
    from bench import windowed_sinc_raw
    fin, fc, fa, fo, df, tp = args
    try:
        eng = A.PolyDec(tp, df, fin, fc, fa, fo, n_channels=2)
        hh = np.concatenate([windowed_sinc_raw(tp * df - 1, 0.4 / df, fc.W - fc.I), [0]])
        eng.set_coeffs(np.array([hh[d + t * df] for d in range(df) for t in range(tp)], dtype=np.int64))
    except A.AcdspError:
        return "rejected"
    eng.run(_buf(fin, 2, (1 << 14) // (16 * df) * (16 * df)))
    return eng.path


MV_TYPES = {"i16": (F(16, 8), F(16, 2), F(40, 18)), "i12": (F(12, 4), F(16, 2), F(30, 10)), "i32": (F(32, 16), F(16, 2), F(56, 30)), "i16_lossy": (F(16, 8), F(16, 2), F(24, 10)),
            "i16_sat": (F(16, 8), F(16, 2), F(40, 18, True, "TRN", "SAT"))}


def mvavg_grid():
    for tname, (fin, fc, fa) in MV_TYPES.items():
        for taps in (3, 9, 17, 33, 65):
            for mode in ("MIRROR", "WIN", "CLIP"):
                for ns in (1024, 1000, 128):
                    if (mode != "MIRROR" and (taps not in (9, 33) or ns != 1024)) or (ns != 1024 and taps != 9):
                        continue
                    for oname, fo in (("o16", F(16, 8, True, "RND", "SAT")), ("o_acc", F(fa.W, fa.I))):
                        yield "mvavg|%s|T%d|%s|ns%d|%s" % (tname, taps, mode, ns, oname), (fin, fc, fa, fo, taps, mode, ns)


def mvavg_resolve(args):
    fin, fc, fa, fo, taps, mode, ns = args
    try:
        eng = A.MvAvg(4096, taps, mode, fin, fc, fa, fo, n_objects=2)
        w = np.hanning(taps + 2)[1:-1]
        eng.set_coeffs(np.round(w / w.sum() * 2.0 ** (fc.W - fc.I)).astype(np.int64))
    except A.AcdspError:
        return "rejected"
    eng.run(_buf(fin, 2, 8 * ns), ns)
    return eng.path


ID_TYPES = {"i16": (F(16, 8), F(32, 16), F(32, 16)), "i16_o16": (F(16, 8), F(32, 16), F(16, 8, True, "RND", "SAT")), "i32": (F(32, 16), F(48, 32), F(48, 32)),
            "i12_sat": (F(12, 4), F(24, 12, True, "TRN", "SAT"), F(24, 12)), "i32_a64": (F(32, 16), F(64, 32), F(64, 32))}


def intgdump_grid():
    for tname, (fin, fa, fo) in ID_TYPES.items():
        for chn in (1, 2, 3, 4, 7, 8, 16):
            for ns in (7, 8, 64, 1000):      # 7: blocks of 7 CHN samples are no multiple of 16 for odd CHN -- the tiled kernel
                yield "intgdump|%s|CHN%d|NS%d" % (tname, chn, ns), (fin, fa, fo, chn, ns)


def intgdump_resolve(args):
    fin, fa, fo, chn, ns = args
    try:
        eng = A.IntgDump(ns, chn, fin, fa, fo, n_objects=2)
    except A.AcdspError:
        return "rejected"
    blocks = max(2, (1 << 14) // (ns * chn))
    eng.run(_buf(fin, 2, blocks * ns * chn), np.full(blocks, ns, dtype=np.int64))
    return eng.path


FAMILIES = {"cic": (cic_grid, cic_resolve), "polydec": (polydec_grid, polydec_resolve), "mvavg": (mvavg_grid, mvavg_resolve), "intgdump": (intgdump_grid, intgdump_resolve)}


def table():
    out = {}
    for gname, (grid, resolve) in FAMILIES.items():
        for k, a in grid():
            out[k] = resolve(a)
    torch.cuda.synchronize()
    return out
