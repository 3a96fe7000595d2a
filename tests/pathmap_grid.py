"""The descriptor grid of tests/test_pathmap_gpu.py and tools/gen_path_map.py: every FIR eligibility predicate of engine.hip / fir_mfma.hip /
fir_gen.hip / fir_generic.hip is crossed at least once on each side (class, ftype, tap count, sample / coefficient / accumulator / output
formats, coefficient sharing, coefficient-set shape)."""
import numpy as np

import ac_dsp_amd as A

F = A.Fmt
TYPES = {
    # name: (IN, COEFF, ACC)
    "i16_exact": (F(16, 2), F(16, 2), F(40, 12)),
    "i16_exact_narrow_acc": (F(16, 2), F(16, 2), F(30, 2)),            # an accumulator the full-scale sum can wrap
    "i16_unsigned_acc": (F(16, 2), F(16, 2), F(40, 12, False)),
    "u16": (F(16, 2, False), F(16, 2), F(44, 16)),
    "u15": (F(15, 3, False), F(16, 2), F(44, 16)),
    "i12": (F(12, 1), F(12, 1), F(31, 8)),
    "i8": (F(8, 1), F(8, 1), F(22, 7)),
    "i16_lossy_trn": (F(16, 2), F(16, 2), F(24, 8)),
    "i16_lossy_rnd_s4": (F(16, 2), F(16, 2), F(38, 14, True, "RND", "WRAP")),
    "i16_sat_acc": (F(16, 2), F(16, 2), F(30, 4, True, "TRN", "SAT")),
    "i16_conv_acc": (F(16, 2), F(16, 2), F(30, 4, True, "RND_CONV", "WRAP")),
    "rtest_const": (F(16, 8), F(32, 16), F(64, 32)),
    "rtest_load": (F(32, 16), F(32, 16), F(64, 32)),
    "rtest_prog": (F(28, 6), F(23, 7), F(64, 32)),
    "prog_rnd": (F(28, 6), F(23, 7), F(64, 32, True, "RND", "WRAP")),
    "prog_s14": (F(28, 6), F(23, 7), F(64, 40)),
    "i32_c16": (F(32, 16), F(16, 2), F(56, 26)),
    "i24_lossy_s4": (F(24, 8), F(20, 4), F(50, 22)),
    "ddc_stage": (F(36, 21), F(16, 1), F(59, 29)),
    "wide_acc": (F(32, 16), F(32, 16), F(80, 40)),
}
OUTS = {
    "o16_rnd_sat": lambda fa: F(16, 2, True, "RND", "SAT"),
    "o16_trn_wrap": lambda fa: F(16, 2, True, "TRN", "WRAP"),
    "o16_conv_satsym": lambda fa: F(16, 2, True, "RND_CONV", "SAT_SYM"),
    "o12_rnd_sat": lambda fa: F(12, 1, True, "RND", "SAT"),
    "o24_rnd_sat": lambda fa: F(24, 6, True, "RND", "SAT"),
    "o_acc": lambda fa: F(fa.W, fa.I, fa.S),
}
TAPS = [1, 27, 28, 63, 130, 255, 300, 1023]
KINDS_FTYPES = [("load", "SHIFT_REG"), ("load", "C_BUFF"), ("prog", "FOLD_ODD"), ("const", "FOLD_EVEN"), ("load", "TRANSPOSED"), ("const", "TRANSPOSED"),
                ("reg_share", "FOLD_ODD_ANTI"), ("reg_share", "SHIFT_REG")]
SETS = ["sinc", "dense", "small"]


def coeffs_of(kind_of_set, n_taps, fc):
    from bench import windowed_sinc_raw
    lim = (1 << (fc.W - 1)) - 1
    if kind_of_set == "sinc":
        return np.clip(windowed_sinc_raw(n_taps | 1, 0.1, fc.W - fc.I)[:n_taps], -lim, lim)
    rng = np.random.default_rng(n_taps)
    c = rng.integers(-lim, lim + 1, size=n_taps, dtype=np.int64)
    return c if kind_of_set == "dense" else c >> max(fc.W - 7, 0)


def grid():
    """yields (key, constructor arguments)"""
    for tname, (fin, fc, fa) in TYPES.items():
        for oname, mk in OUTS.items():
            fo = mk(fa)
            for n_taps in TAPS:
                for kind, ftype in KINDS_FTYPES:
                    if kind == "reg_share" and (fa.W > 64 or fo.W > 64):
                        continue
                    nt = n_taps if not (ftype == "FOLD_EVEN" and n_taps % 2) else n_taps + 1
                    for sname in SETS:
                        if sname != "sinc" and (n_taps not in (63, 255, 1023) or ftype != "SHIFT_REG"):
                            continue
                        for per_ch in ((False, True) if (ftype == "SHIFT_REG" and kind == "load" and n_taps in (63, 300)) else (False,)):
                            key = "%s|%s|%d|%s|%s|%s|%d" % (tname, oname, nt, kind, ftype, sname, int(per_ch))
                            yield key, (nt, ftype, fin, fc, fa, fo, kind, per_ch, sname)


def resolve(args):
    nt, ftype, fin, fc, fa, fo, kind, per_ch, sname = args
    try:
        fir = A.Fir(nt, ftype, fin, fc, fa, fo, n_channels=2, kind=kind, coeffs_per_channel=per_ch)
    except A.AcdspError as e:
        return "rejected"
    c = coeffs_of(sname, nt, fc)
    try:
        fir.set_coeffs(np.stack([c, c]) if per_ch else c)
    except A.AcdspError:
        return "rejected_set"
    e = fir.mfma_epilogue()
    return "%s/%d" % (fir.kernel, fir.mfma_issued()) + ("" if e is None else "/e%d%s%s" % (e[0], "+c%d" % e[1] if e[1] else "", "+flip" if e[2] else ""))
