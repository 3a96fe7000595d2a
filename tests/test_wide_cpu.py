"""Types wider than 64 bits, CPU side: oracle/acdsp_oracle_wide.cpp against (a) oracle/acdsp_oracle.c on formats of <= 64 bits, where
the C oracle is pinned by the reference's vectors, (b) the pure-Python big-integer model of tests/wide_model.py at 72 - 128 bits, and
(c) the conversion rules one by one.  (Reference: INT_TYPE of any width, ac_cic_dec_full.h:116-137; ACC_TYPE of any width,
ac_fir_const_coeffs.h:190-296.)"""
import numpy as np
import pytest

import wide_model as M
from oracle import Fmt, OracleCic, OracleCicW, OracleFir, OracleFirW, cic_int_type, requant, requant_wide

FT = ["SHIFT_REG", "ROTATE_SHIFT", "C_BUFF", "FOLD_EVEN", "FOLD_ODD", "TRANSPOSED"]
QS = list(M.Q)
OS = list(M.O)


def ofmt(f):
    return Fmt(f.W, f.I, f.S, f.Q, f.O)


def rnd_raw(rng, f, size):
    return np.array([int(rng.integers(f.lo >> 32, (f.hi >> 32) + 1)) * (1 << 32) + int(rng.integers(0, 1 << 32)) if f.W > 40
                     else int(rng.integers(f.lo, f.hi + 1)) for _ in range(size)], dtype=object).clip(f.lo, f.hi)


def test_requant_wide_agrees_with_the_model_on_every_mode():
    rng = np.random.default_rng(1)
    for _ in range(4000):
        W = int(rng.integers(1, 129))
        S = bool(rng.integers(0, 2)) if W not in (64, 128) else True
        d = M.F(W, int(rng.integers(-20, W + 20)), S, QS[rng.integers(0, 8)], OS[rng.integers(0, 4)])
        bits = int(rng.integers(1, 200))
        x = int(rng.integers(-2 ** 62, 2 ** 62)) * (1 << max(bits - 62, 0)) + int(rng.integers(0, 2 ** 30))
        f_src = int(rng.integers(d.F - 60, d.F + 120))
        assert requant_wide(x, f_src, ofmt(d)) == M.requant(x, f_src, d), (x, f_src, vars(d))


def test_requant_wide_agrees_with_the_c_oracle_at_64_bits():
    rng = np.random.default_rng(2)
    for _ in range(3000):
        W = int(rng.integers(1, 64))
        d = Fmt(W, int(rng.integers(-10, W + 10)), bool(rng.integers(0, 2)), int(rng.integers(0, 8)), int(rng.integers(0, 4)))
        x = int(rng.integers(-2 ** 62, 2 ** 62)) * int(rng.integers(1, 2 ** 40))
        f_src = int(rng.integers(d.F - 20, d.F + 90))
        assert requant_wide(x, f_src, d) == requant(x, f_src, d)


@pytest.mark.parametrize("ftype", FT)
def test_wide_fir_oracle_equals_the_c_oracle_on_narrow_formats(ftype):
    rng = np.random.default_rng(hash(ftype) % 1000)
    for case in range(12):
        fin = Fmt(int(rng.integers(4, 33)), int(rng.integers(1, 12)), True)
        fc = Fmt(int(rng.integers(4, 24)), int(rng.integers(1, 8)), bool(rng.integers(0, 2)))
        fa = Fmt(int(rng.integers(20, 63)), int(rng.integers(8, 30)), True, int(rng.integers(0, 8)), int(rng.integers(0, 4)))
        fo = Fmt(int(rng.integers(8, 63)), int(rng.integers(4, 20)), True, int(rng.integers(0, 8)), int(rng.integers(0, 4)))
        n_taps = int(rng.integers(1, 24))
        c = rng.integers(-(1 << (fc.W - 1)) if fc.S else 0, (1 << (fc.W - 1)) if fc.S else (1 << fc.W), size=n_taps, dtype=np.int64)
        x = rng.integers(-(1 << (fin.W - 1)), 1 << (fin.W - 1), size=(2, 80), dtype=np.int64)
        try:
            a = OracleFir(n_taps, ftype, fin, fc, fa, fo, n_ch=2).run(c, x)
        except ValueError:
            continue
        b = OracleFirW(n_taps, ftype, fin, fc, fa, fo, n_ch=2).run(c, x)
        assert np.array_equal(a.astype(object), b), (case, ftype)


@pytest.mark.parametrize("interp", [0, 1])
def test_wide_cic_oracle_equals_the_c_oracle_on_narrow_formats(interp):
    rng = np.random.default_rng(7 + interp)
    for R, Mm, N, W in ((8, 1, 5, 32), (7, 2, 4, 32), (3, 3, 3, 16), (16, 1, 5, 16), (4, 4, 2, 24)):
        fin = Fmt(W, W // 2, True)
        it = cic_int_type(interp, R, Mm, N, fin)
        for fo in (Fmt(it.W, it.I), Fmt(20, 9, True, "RND", "SAT"), Fmt(33, 20, False, "TRN_ZERO", "SAT_SYM")):
            a, b = OracleCic(interp, R, Mm, N, fin, fo, n_ch=2), OracleCicW(interp, R, Mm, N, fin, fo, n_ch=2)
            for n in (37, 1, 64):      # several run() calls: state carries in both
                x = rng.integers(-(1 << (W - 1)), 1 << (W - 1), size=(2, n), dtype=np.int64)
                assert np.array_equal(a.run(x).astype(object), b.run(x)), (interp, R, Mm, N, n)


WIDE_FIR = [
    # IN, COEFF, ACC, OUT
    (M.F(32, 16), M.F(32, 16), M.F(96, 48), M.F(96, 48)),                               # the reference testbench types with a 96-bit ACC
    (M.F(64, 32), M.F(32, 8), M.F(128, 60), M.F(100, 50, True, "RND", "SAT")),
    (M.F(16, 2), M.F(16, 2), M.F(72, 44, True, "RND_CONV", "SAT_SYM"), M.F(72, 44)),    # lossy never; saturating 72-bit ACC
    (M.F(40, 20), M.F(24, 4), M.F(80, 30, True, "TRN_ZERO", "WRAP"), M.F(66, 30, True, "RND_INF", "SAT_ZERO")),   # lossy ACC: per-tap order matters
    (M.F(20, 10, False), M.F(18, 4), M.F(90, 50, False, "RND", "SAT"), M.F(70, 40, False)),                        # unsigned chain
]


@pytest.mark.parametrize("ftype", FT)
@pytest.mark.parametrize("k", range(len(WIDE_FIR)))
def test_wide_fir_oracle_equals_the_python_model(ftype, k):
    fin, fc, fa, fo = WIDE_FIR[k]
    rng = np.random.default_rng(100 * k + FT.index(ftype))
    for n_taps in (1, 6, 9):
        c = [int(v) for v in rnd_raw(rng, fc, n_taps)]
        x = [int(v) for v in rnd_raw(rng, fin, 40)]
        want = M.fir(ftype, c, x, fin, fc, fa, fo)
        got = OracleFirW(n_taps, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo)).run(np.array(c, dtype=np.int64), np.array([x], dtype=np.int64))[0]
        assert list(got) == want, (ftype, k, n_taps)


# (the reference's power<> is an int enum, ac_cic_dec_full.h:94-105: (R M)^N must stay below 2^31, so INT_TYPE exceeds 64 bits only for
# inputs of more than 33 bits -- e.g. <48,20> through R 16, M 2, N 6 is 78 bits)
WIDE_CIC = [(0, 16, 2, 6, M.F(48, 20)), (1, 16, 2, 6, M.F(48, 20)), (0, 8, 1, 8, M.F(45, 20)), (1, 8, 3, 5, M.F(64, 30)), (0, 255, 1, 3, M.F(60, 30, False))]


@pytest.mark.parametrize("interp,R,Mm,N,fin", WIDE_CIC)
def test_wide_cic_oracle_equals_the_python_model(interp, R, Mm, N, fin):
    it = M.cic_int_type(interp, R, Mm, N, fin)
    assert it.W > 64
    oit = cic_int_type(interp, R, Mm, N, ofmt(fin))
    assert (oit.W, oit.I) == (it.W, it.I)
    rng = np.random.default_rng(R * N)
    for fo in (it, M.F(it.W - 7, it.I - 3, True, "RND", "SAT"), M.F(40, 20, True, "RND_CONV", "SAT")):
        o = OracleCicW(interp, R, Mm, N, ofmt(fin), ofmt(fo))
        st = None
        for n in ((3 * R + 5, 2, R) if not interp else (9, 1, 4)):
            x = [int(v) for v in rnd_raw(rng, fin, n)]
            want, st = M.cic(interp, R, Mm, N, x, fin, fo, st)
            got = o.run(np.array([x], dtype=np.int64))
            assert (list(got[0]) if got.size else []) == want, (interp, R, Mm, N, n)


def wide_cases(kind):
    import golden_cases as G
    return G.load((kind,))


@pytest.mark.parametrize("c", wide_cases("wide_fir"), ids=lambda c: c["name"])
def test_wide_fir_oracle_matches_the_reference_headers_vectors(c):
    """tests/golden/ref_hdr/wide.json: the reference's own ac_fir_load_coeffs source compiled over include/ac_types with an 80-bit
    accumulator (tools/gen_golden/gen_wide.cpp) -- pins loop order, the ACC-typed fold and the per-tap conversions at this width."""
    f = [Fmt(*c[k][:2], bool(c[k][2]), c[k][3], c[k][4]) for k in ("in", "coeff", "acc", "out")]
    o = OracleFirW(c["n_taps"], c["ftype"], *f)
    x = np.array(c["x"], dtype=np.int64)
    got, pos = [], 0
    for k in c["calls"]:
        got += list(o.run(np.array(c["coeffs"], dtype=np.int64), x[None, pos:pos + k])[0])
        pos += k
    assert got == c["y"]


@pytest.mark.parametrize("c", wide_cases("wide_cic_dec") + wide_cases("wide_cic_intr"), ids=lambda c: c["name"])
def test_wide_cic_oracle_matches_the_reference_headers_vectors(c):
    fin, fo = (Fmt(*c[k][:2], bool(c[k][2]), c[k][3], c[k][4]) for k in ("in", "out"))
    o = OracleCicW(int(c["class"] == "wide_cic_intr"), c["R"], c["M"], c["N"], fin, fo)
    x = np.array(c["x"], dtype=np.int64)
    got, pos = [], 0
    for k, want_n in zip(c["calls"], c["outs_per_call"]):
        y = o.run(x[None, pos:pos + k])
        assert (y.shape[1] if y.size else 0) == want_n
        got += list(y[0]) if y.size else []
        pos += k
    assert got == c["y"]
