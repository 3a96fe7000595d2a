"""CPU tests of the oracle itself: pinned on the reference's own vectors, on hand-derived known answers
for every quantisation / overflow mode, and cross-checked against the independent ac_fixed templates."""
import os
import subprocess

import numpy as np
import pytest

from oracle import Fmt, OracleFir, OracleCic, requant, from_double, stimulus, cic_int_type, lib
from helpers import read_fracs, to_raw, two_tone, sqnr_db

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ---------------------------------------------------------------- pins on the reference's vectors


def test_cic_decimator_reference_vector_exact():
    x = np.concatenate([[0], to_raw(read_fracs("ac_cic_dec_full_input.txt"), 16)])
    ref = to_raw(read_fracs("ac_cic_dec_full_ref.txt"), 16)
    y = OracleCic(0, 7, 2, 4, Fmt(32, 16), Fmt(48, 32)).run(x)[0]
    assert len(x) == 10004 and len(y) == 1430 and len(ref) == 1429
    assert np.array_equal(y[:1429], ref)


def test_cic_interpolator_reference_vector_exact():
    x = to_raw(read_fracs("ac_cic_intr_full_input.txt"), 16)[:1000]
    ref = to_raw(read_fracs("ac_cic_intr_full_ref.txt"), 16)[5:]
    y = OracleCic(1, 7, 2, 5, Fmt(32, 16), Fmt(49, 33)).run(x)[0]
    assert len(y) == 6990 and np.array_equal(y, ref[:6990])


@pytest.mark.parametrize("name,taps,fi,fc,want", [("const", 29, Fmt(16, 8), Fmt(32, 16), 84.24),
                                                  ("load", 27, Fmt(32, 16), Fmt(32, 16), 89.56),
                                                  ("prog", 27, Fmt(28, 6), Fmt(23, 7), 89.56)])
def test_fir_reference_vectors_sqnr(name, taps, fi, fc, want):
    fa = Fmt(64, 32)
    cfg = read_fracs("ac_fir_%s_coeffs_cfg.txt" % name)
    assert len(cfg) == taps
    c = np.array([from_double(float(v), fc) for v in cfg], dtype=np.int64)
    assert all(int(v * (1 << fc.F)) == r for v, r in zip(cfg, c))   # coefficients are exactly representable
    y = OracleFir(taps, "FOLD_ODD", fi, fc, fa, fa).run(c, two_tone(fi))[0]
    got = sqnr_db(y, 32, read_fracs("ac_fir_%s_coeffs_ref.txt" % name)[:1024])
    assert got >= 60.0 and abs(got - want) < 0.01


def test_cic_intermediate_type_matches_reference_params():
    assert (cic_int_type(0, 7, 2, 4, Fmt(32, 16)).W, cic_int_type(0, 7, 2, 4, Fmt(32, 16)).I) == (48, 32)
    assert cic_int_type(1, 7, 2, 5, Fmt(32, 16)).W == 49
    assert cic_int_type(0, 8, 1, 5, Fmt(32, 16)).W == 47
    assert cic_int_type(0, 16, 1, 5, Fmt(16, 1)).W == 36
    assert cic_int_type(0, 8, 1, 5, Fmt(12, 4, False)).W == 28   # unsigned input: one extra bit
    with pytest.raises(ValueError):
        cic_int_type(0, 64, 4, 8, Fmt(32, 16))                  # int power<> overflow in the reference

# ---------------------------------------------------------------- known answers, hand-derived


def q(x, fsrc, W, I, S, Q, O):
    return requant(x, fsrc, Fmt(W, I, S, Q, O))


@pytest.mark.parametrize("mode,expected", [
    # dropping 2 fractional bits of raw x/4:   x =  -7   -6   -5   -3   -2   -1    1    2    3    5    6    7
    ("TRN",          [-2, -2, -2, -1, -1, -1, 0, 0, 0, 1, 1, 1]),
    ("RND",          [-2, -1, -1, -1,  0,  0, 0, 1, 1, 1, 2, 2]),
    ("TRN_ZERO",     [-1, -1, -1,  0,  0,  0, 0, 0, 0, 1, 1, 1]),
    ("RND_ZERO",     [-2, -1, -1, -1,  0,  0, 0, 0, 1, 1, 1, 2]),
    ("RND_INF",      [-2, -2, -1, -1, -1,  0, 0, 1, 1, 1, 2, 2]),
    ("RND_MIN_INF",  [-2, -2, -1, -1, -1,  0, 0, 0, 1, 1, 1, 2]),
    ("RND_CONV",     [-2, -2, -1, -1,  0,  0, 0, 0, 1, 1, 2, 2]),
    ("RND_CONV_ODD", [-2, -1, -1, -1, -1,  0, 0, 1, 1, 1, 1, 2]),
])
def test_quantisation_modes_known_answers(mode, expected):
    xs = [-7, -6, -5, -3, -2, -1, 1, 2, 3, 5, 6, 7]
    got = [q(x, 2, 16, 16, True, mode, "WRAP") for x in xs]
    assert got == expected


def test_overflow_modes_known_answers():
    # 4-bit signed destination: range [-8, 7]
    assert [q(x, 0, 4, 4, True, "TRN", "WRAP") for x in (7, 8, 9, -8, -9, 23)] == [7, -8, -7, -8, 7, 7]
    assert [q(x, 0, 4, 4, True, "TRN", "SAT") for x in (7, 8, 100, -8, -9, -100)] == [7, 7, 7, -8, -8, -8]
    assert [q(x, 0, 4, 4, True, "TRN", "SAT_ZERO") for x in (7, 8, -8, -9)] == [7, 0, -8, 0]
    assert [q(x, 0, 4, 4, True, "TRN", "SAT_SYM") for x in (7, 8, -7, -8, -9)] == [7, 7, -7, -7, -7]
    # unsigned 4-bit: range [0, 15]
    assert [q(x, 0, 4, 4, False, "TRN", "WRAP") for x in (15, 16, -1)] == [15, 0, 15]
    assert [q(x, 0, 4, 4, False, "TRN", "SAT") for x in (15, 16, -1)] == [15, 15, 0]
    assert [q(x, 0, 4, 4, False, "TRN", "SAT_ZERO") for x in (15, 16, -1)] == [15, 0, 0]
    # the rounding carry takes part in the overflow decision: 7.75 -> RND -> 8 -> saturates to 7
    assert q(31, 2, 4, 4, True, "RND", "SAT") == 7
    assert q(31, 2, 4, 4, True, "RND", "WRAP") == -8


def test_double_to_fixed():
    f = Fmt(16, 8)
    assert from_double(1.5, f) == 384 and from_double(-2.25, f) == -576
    assert from_double(-0.001, f) == -1 and from_double(0.001, f) == 0          # AC_TRN = floor
    assert from_double(127.99609375, f) == 32767
    assert from_double(-0.001, Fmt(16, 8, True, "TRN_ZERO")) == 0
    assert from_double(7.99, Fmt(8, 4, True, "RND", "SAT")) == 127


def test_comb_delay_line_quirk_m_ge_3_equals_m_2():
    # ac_cic_full_core.h:249-254 shifts the delay line in ascending order: for M >= 3 the effective
    # differential delay is 2.  Output TYPE still depends on M, so compare with a matching wide OUT.
    x = stimulus(5, 1, 400, 16)
    fin, fout = Fmt(16, 4), Fmt(40, 28)
    y2 = OracleCic(0, 3, 2, 3, fin, fout).run(x)
    y3 = OracleCic(0, 3, 3, 3, fin, fout).run(x)
    y4 = OracleCic(0, 3, 4, 3, fin, fout).run(x)
    assert np.array_equal(y2, y3) and np.array_equal(y2, y4)


def test_cic_closed_form_fir_identity():
    # dec output m = sum_k h[k] x[mR-(N-1)-k] mod 2^W, h = boxcar(R*M)^N   -- the identity the
    # GPU kernel's warm-up argument rests on
    R, M, N = 5, 2, 3
    fin = Fmt(20, 8)
    it = cic_int_type(0, R, M, N, fin)
    x = stimulus(11, 1, 600, 20)[0]
    y = OracleCic(0, R, M, N, fin, Fmt(it.W, it.I)).run(x)[0]
    h = np.array([1], dtype=object)
    for _ in range(N):
        h = np.convolve(h, np.ones(R * M, dtype=object))
    xp = np.concatenate([np.zeros(len(h) + N, dtype=object), x.astype(object)])
    for m in range(len(y)):
        t = m * R - (N - 1) + len(h) + N
        acc = sum(int(h[k]) * int(xp[t - k]) for k in range(len(h)))
        acc &= (1 << it.W) - 1
        if acc >= 1 << (it.W - 1):
            acc -= 1 << it.W
        assert acc == y[m], m


def test_fir_architectures_agree_when_lossless_and_symmetric():
    fi, fc, fa = Fmt(16, 2), Fmt(16, 2), Fmt(40, 12)
    rng = np.random.default_rng(0)
    half = rng.integers(-3000, 3000, size=16)
    c = np.concatenate([half, [1234], half[::-1]]).astype(np.int64)     # 33 taps, symmetric
    x = stimulus(1, 1, 500, 16)
    ys = [OracleFir(33, ft, fi, fc, fa, fa).run(c, x) for ft in ("SHIFT_REG", "ROTATE_SHIFT", "C_BUFF", "FOLD_ODD", "TRANSPOSED")]
    for y in ys[1:]:
        assert np.array_equal(ys[0], y)
    direct = np.convolve(x[0].astype(object), c.astype(object))[:500]
    assert all(int(a) == int(b) for a, b in zip(ys[0][0], direct))


def test_fir_state_carries_across_calls():
    fi, fc, fa = Fmt(28, 6), Fmt(23, 7), Fmt(64, 32)
    rng = np.random.default_rng(2)
    c = rng.integers(-2 ** 20, 2 ** 20, size=27)
    x = stimulus(3, 1, 300, 28)
    for ft in ("SHIFT_REG", "C_BUFF", "FOLD_EVEN", "FOLD_ODD", "TRANSPOSED"):
        whole = OracleFir(27, ft, fi, fc, fa, fa).run(c, x)
        o = OracleFir(27, ft, fi, fc, fa, fa)
        parts = np.concatenate([o.run(c, x[:, :1]), o.run(c, x[:, 1:100]), o.run(c, x[:, 100:])], axis=1)
        assert np.array_equal(whole, parts), ft


def test_anti_ftypes_unhandled_like_reference():
    with pytest.raises(ValueError):
        OracleFir(8, "FOLD_EVEN_ANTI", Fmt(16, 2), Fmt(16, 2), Fmt(40, 12), Fmt(16, 2)).run(np.zeros(8, dtype=np.int64), np.zeros((1, 4), dtype=np.int64))

# ---------------------------------------------------------------- cross-check with the ac_fixed templates


FM = {
    # case id -> (src fmt, dst fmt) for C lines;  (a, b, acc) for M lines
}


def _case_formats():
    qn = ["TRN", "RND", "TRN_ZERO", "RND_ZERO", "RND_INF", "RND_MIN_INF", "RND_CONV", "RND_CONV_ODD"]
    on = ["WRAP", "SAT", "SAT_ZERO", "SAT_SYM"]
    conv, mac = {}, {}
    for qi, qm in enumerate(qn):
        for oi, om in enumerate(on):
            conv[100 + qi * 4 + oi] = (Fmt(24, 9, True), Fmt(12, 5, True, qm, om))
            conv[200 + qi * 4 + oi] = (Fmt(24, 9, True), Fmt(11, 4, False, qm, om))
            mac[400 + qi * 4 + oi] = (Fmt(12, 4, True), Fmt(10, 2, True), Fmt(18, 7, True, qm, om))
        conv[300 + qi * 4 + 1] = (Fmt(20, 6, False), Fmt(10, 3, True, qm, "SAT"))
        conv[300 + qi * 4 + 3] = (Fmt(20, 6, False), Fmt(10, 8, True, qm, "SAT_SYM"))
    conv[900] = (Fmt(16, 8), Fmt(64, 32))
    conv[901] = (Fmt(28, 6), Fmt(64, 32))
    mac[902] = (Fmt(28, 6), Fmt(23, 7), Fmt(64, 32))
    mac[903] = (Fmt(64, 32), Fmt(32, 16), Fmt(64, 32))
    return conv, mac


def test_oracle_agrees_with_ac_fixed_templates(tmp_path):
    exe = str(tmp_path / "xcheck")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I" + os.path.join(ROOT, "include", "ac_types"),
                           os.path.join(ROOT, "tests", "cpp", "xcheck_fixed.cpp"), "-o", exe])
    out = subprocess.check_output([exe]).decode().split("\n")
    conv, mac = _case_formats()
    nc = nm = 0
    for line in out:
        p = line.split()
        if not p:
            continue
        if p[0] == "C":
            src, dst = conv[int(p[1])]
            assert requant(int(p[2]), src.F, dst) == int(p[3]), line
            nc += 1
        else:
            a, b, acc = mac[int(p[1])]
            f = max(acc.F, a.F + b.F)
            s = (int(p[2]) << (f - acc.F)) + ((int(p[3]) * int(p[4])) << (f - a.F - b.F))
            assert requant(s, f, acc) == int(p[5]), line
            nm += 1
    assert nc > 20000 and nm > 9000


def test_numpy_stimulus_equals_c_stimulus():
    a = stimulus(0xACD5, 3, 50, 16, ch0=2, t0=7)
    for c in range(3):
        for t in range(50):
            assert a[c, t] == lib.orc_stimulus(0xACD5, 2 + c, 7 + t, 16)


def test_polydec_is_a_decimating_fir_when_lossless():
    # ac_poly_dec.h:112-128: y[g] = sum_k h[k] x[g*DF + DF-1 - k] with h[df + tp*DF] = c[tp + NTAPS*df]
    from oracle import OraclePolyDec
    NT, DF = 5, 3
    fi, fc, fa = Fmt(16, 2), Fmt(16, 2), Fmt(40, 12)
    rng = np.random.default_rng(0)
    c = rng.integers(-2000, 2000, size=NT * DF)
    x = stimulus(3, 1, 302, 16)                     # 100 groups + 2 left-over samples (not consumed)
    y = OraclePolyDec(NT, DF, fi, fc, fa, fa).run(c, x)[0]
    assert len(y) == 100
    h = [0] * (NT * DF)
    for df in range(DF):
        for tp in range(NT):
            h[df + tp * DF] = int(c[tp + NT * df])
    xp = [0] * (NT * DF) + [int(v) for v in x[0]]
    for g in range(100):
        t = g * DF + DF - 1 + NT * DF
        assert y[g] == sum(h[k] * xp[t - k] for k in range(NT * DF))
    # state carries across calls in whole groups
    o = OraclePolyDec(NT, DF, fi, fc, fa, fa)
    parts = np.concatenate([o.run(c, x[:, :30]), o.run(c, x[:, 30:300])], axis=1)
    assert np.array_equal(parts[0], y)


# ---- ac_fir_reg_share ----

def _rs(ftype, n, fin, fc, fa, fo, **kw):
    return OracleFir(n, ftype, fin, fc, fa, fo, reg_share=kw.get("reg_share", (1, 1, 0)))


def test_reg_share_impulse_responses_and_delay_line():
    # known answers by hand: an impulse reads the effective tap sequence back; the anti-symmetric folds negate the
    # mirrored half (ac_fir_reg_share.h:178-192, 226-246), FOLD_ODD's centre tap passes through once (:207-208)
    f, a = Fmt(16, 2), Fmt(40, 12)
    imp = np.zeros((1, 12), dtype=np.int64)
    imp[0, 0] = 1 << 14                                  # 1.0 in <16,2>
    c8 = np.array([1, 2, 3, 4, 5, 6, 7, 8], dtype=np.int64) << 14
    c7 = c8[:7]
    sc = lambda v: [int(t) << 28 for t in v]             # products carry 28 fraction bits; OUT = ACC keeps them
    assert _rs("SHIFT_REG", 8, f, f, a, a).run(c8, imp)[0, :9].tolist() == sc([1, 2, 3, 4, 5, 6, 7, 8, 0])
    assert _rs("FOLD_EVEN", 8, f, f, a, a).run(c8, imp)[0, :9].tolist() == sc([1, 2, 3, 4, 4, 3, 2, 1, 0])
    assert _rs("FOLD_EVEN_ANTI", 8, f, f, a, a).run(c8, imp)[0, :9].tolist() == sc([1, 2, 3, 4, -4, -3, -2, -1, 0])
    assert _rs("FOLD_ODD", 7, f, f, a, a).run(c7, imp)[0, :8].tolist() == sc([1, 2, 3, 4, 3, 2, 1, 0])
    o = _rs("FOLD_ODD_ANTI", 7, f, f, a, a)
    assert o.run(c7, imp)[0, :8].tolist() == sc([1, 2, 3, 4, -3, -2, -1, 0])
    # delay line: reg[N-1] (:128-130) -- the impulse has left the 7-deep register after 12 samples
    assert o.delay_line().tolist() == [0]
    o2 = _rs("SHIFT_REG", 8, f, f, a, Fmt(16, 2))
    o2.run(c8, imp[:, :8])
    assert o2.delay_line().tolist() == [1 << 14]         # the impulse now sits in reg[7]


def test_reg_share_blocked_coefficient_memory():
    # tap t reads coeffs[(t / BLK_SZ) * MEM_WORD_WIDTH + BLK_OFFSET + t % BLK_SZ]  (:141-147)
    f, a = Fmt(16, 2), Fmt(40, 12)
    rng = np.random.default_rng(5)
    x = rng.integers(-32768, 32768, size=(1, 64))
    mem = rng.integers(-3000, 3000, size=16)
    # FOLD_EVEN, 16 taps: 8 MACs in 4 blocks of 2, words 4 apart, offset 1 -> addresses 1,2, 5,6, 9,10, 13,14
    tap = np.zeros(16, dtype=np.int64)
    tap[:8] = mem[[1, 2, 5, 6, 9, 10, 13, 14]]
    y_blk = OracleFir(16, "FOLD_EVEN", f, f, a, a, reg_share=(4, 2, 1)).run(mem, x)
    y_lin = OracleFir(16, "FOLD_EVEN", f, f, a, a, reg_share=(1, 1, 0)).run(tap, x)
    assert np.array_equal(y_blk, y_lin)
    # ... and the symmetric fold equals the const/load/prog FOLD_EVEN core on a lossless accumulator
    assert np.array_equal(y_lin, OracleFir(16, "FOLD_EVEN", f, f, a, a).run(tap, x))
    # where the reference would index outside its arrays / has no branch: refused
    for bad in [dict(n=12, ft="SHIFT_REG", rs=(4, 5, 0)), dict(n=12, ft="SHIFT_REG", rs=(4, 3, 1)), dict(n=8, ft="C_BUFF", rs=(1, 1, 0))]:
        with pytest.raises(ValueError):
            OracleFir(bad["n"], bad["ft"], f, f, a, a, reg_share=bad["rs"]).run(np.zeros(bad["n"], dtype=np.int64), x)


def test_reg_share_ascending_order_differs_from_prog_coeffs_only_when_the_accumulator_saturates():
    f = Fmt(16, 2)
    rng = np.random.default_rng(6)
    x = rng.integers(-32768, 32768, size=(1, 200))
    c = rng.integers(-32768, 32768, size=9)
    wide = Fmt(40, 12)
    assert np.array_equal(OracleFir(9, "SHIFT_REG", f, f, wide, wide, reg_share=(1, 1, 0)).run(c, x),
                          OracleFir(9, "SHIFT_REG", f, f, wide, wide).run(c, x))
    sat = Fmt(30, 2, True, "TRN", "SAT")                 # saturates on the way: i = 0..N-1 vs i = N-1..0 give different sums
    y_rs = OracleFir(9, "SHIFT_REG", f, f, sat, wide, reg_share=(1, 1, 0)).run(c, x)
    y_pg = OracleFir(9, "SHIFT_REG", f, f, sat, wide).run(c, x)
    assert not np.array_equal(y_rs, y_pg)
    assert np.array_equal(y_rs, OracleFir(9, "C_BUFF", f, f, sat, wide).run(c, x))   # C_BUFF is the ascending const/load/prog core


# ---- ac_poly_intr (second half) ----

def test_poly_intr_known_answers():
    from oracle import OraclePolyIntr
    f, a = Fmt(16, 2), Fmt(40, 12)
    one = 1 << 14
    x = np.zeros((1, 8), dtype=np.int64)
    x[0, 0] = one                                        # unit impulse
    # FOLD_ANTI is the plain polyphase MAC (ac_poly_intr.h:246-256): phase j of sample n = sum_i taps[i] * c[i + N*j],
    # written at once -> the impulse reads the IF x NTAPS coefficient table back column by column
    N, IF = 3, 2
    c = (np.arange(1, N * IF + 1, dtype=np.int64)) << 14
    y = OraclePolyIntr(N, N * IF, IF, "FOLD_ANTI", f, f, a, a).run(c, [1, 1], [0, 1], x)[0]
    sc = lambda v: [int(t) << 28 for t in v]
    assert y[:8].tolist() == sc([1, 4, 2, 5, 3, 6, 0, 0])
    # FOLD_EVEN, corr[j] = j: sum_i c[i + (j*N)/2] * (taps[i] + taps[N-1-i]), emitted ONE SAMPLE LATER (:153-175): the first
    # sample produces nothing, so 8 inputs give 7 * IF outputs; N = 4 -> symmetric impulse response c0 c1 c1 c0 per phase
    N, IF = 4, 2
    c = np.array([1, 2, 5, 7], dtype=np.int64) << 14
    y = OraclePolyIntr(N, 4, IF, "FOLD_EVEN", f, f, a, a).run(c, [1, 1], [0, 1], x)[0]
    assert len(y) == 7 * IF
    assert y[0::2][:5].tolist() == sc([1, 2, 2, 1, 0]) and y[1::2][:5].tolist() == sc([5, 7, 7, 5, 0])
    # sign[j] = 0: the mirrored tap is negated (anti-symmetric phase)
    y = OraclePolyIntr(N, 4, IF, "FOLD_EVEN", f, f, a, a).run(c, [0, 1], [0, 1], x)[0]
    assert y[0::2][:5].tolist() == sc([1, 2, -2, -1, 0])
    # symmetric-pair technique: corr = [1, 0], sign = [1, 1] -> out_j = (acc_j - acc_corr(j)) >> 1   (tn = -t2 when sign[j])
    y = OraclePolyIntr(N, 4, IF, "FOLD_EVEN", f, f, a, a).run(c, [1, 1], [1, 0], x)[0]
    assert y[0::2][:4].tolist() == [(p - q) >> 1 for p, q in zip(sc([1, 2, 2, 1]), sc([5, 7, 7, 5]))]
    # FOLD_ODD: centre tap passes through once, coefficient stride N/2 + 1 per phase (:207)
    N, IF = 5, 2
    c = np.array([1, 2, 3, 4, 5, 6], dtype=np.int64) << 14
    y = OraclePolyIntr(N, 6, IF, "FOLD_ODD", f, f, a, a).run(c, [1, 1], [0, 1], x)[0]
    assert y[0::2][:6].tolist() == sc([1, 2, 3, 2, 1, 0]) and y[1::2][:6].tolist() == sc([4, 5, 6, 5, 4, 0])
    # out-of-range table / bank indices are refused (the reference would read outside its arrays)
    with pytest.raises(ValueError):
        OraclePolyIntr(4, 3, 2, "FOLD_EVEN", f, f, a, a).run(c[:3], [1, 1], [0, 1], x)
    with pytest.raises(ValueError):
        OraclePolyIntr(4, 4, 2, "FOLD_EVEN", f, f, a, a).run(c[:4], [1, 1], [0, 2], x)


# ---- ac_intg_dump ----

def test_intg_dump_known_answers():
    from oracle import OracleIntgDump
    f, a = Fmt(16, 8), Fmt(32, 24)                       # same fraction (8 bits): raw words add as integers
    o = OracleIntgDump(4, 2, f, a, a)                    # NS = 4, two interleaved channels
    # block 1: n_sample = 3 -> three rounds, sums 1+3+5 / 2+4+6; block 2: n_sample = 1 -> one round
    x = np.array([[1, 2, 3, 4, 5, 6, 10, 20]], dtype=np.int64)
    assert o.run(x, [3, 1])[0].tolist() == [9, 12, 10, 20]
    # n_sample = 0 and n_sample > NS: NS rounds, nothing written, the sums carry into the next block (:138-146)
    x = np.arange(1, 2 * (4 + 4 + 2) + 1, dtype=np.int64)[None, :]
    y = o.run(x, [0, 9, 2])[0]
    ch0, ch1 = x[0, 0::2], x[0, 1::2]
    assert y.tolist() == [int(ch0.sum()), int(ch1.sum())] and o.temp.tolist() == [[0, 0]]
    # the undumped sum also survives the end of a call
    o.run(np.array([[1, 1, 2, 2, 3, 3, 4, 4]], dtype=np.int64), [7])
    assert o.temp.tolist() == [[10, 10]]
    assert o.run(np.array([[5, 6]], dtype=np.int64), [1])[0].tolist() == [15, 16]
    # every add is an ACC_TYPE assignment: a saturating accumulator clips on the way, a wrapping one does not
    sat, wrap = Fmt(8, 8, True, "TRN", "SAT"), Fmt(8, 8, True, "TRN", "WRAP")
    fi = Fmt(8, 8)
    x = np.array([[100, 100, -100, -100]], dtype=np.int64)
    assert OracleIntgDump(8, 1, fi, sat, Fmt(16, 16)).run(x, [4])[0].tolist() == [127 - 200]
    assert OracleIntgDump(8, 1, fi, wrap, Fmt(16, 16)).run(x, [4])[0].tolist() == [0]
