"""GPU parity tests, polyphase interpolator (second half): acdsp_polyintr_* vs the oracle restatement of
reference include/ac_dsp/ac_poly_intr.h:104-320 (the reference ships no test or vector for this class)."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from oracle import OraclePolyIntr
from helpers import ofmt

pytestmark = pytest.mark.gpu


def rand_raw(rng, fmt, shape):
    lo = -(1 << (fmt.W - 1)) if fmt.S else 0
    hi = (1 << (fmt.W - 1)) - 1 if fmt.S else (1 << fmt.W) - 1
    return rng.integers(lo, hi + 1, size=shape, dtype=np.int64)


def table_size(n_taps, ifac, ftype):
    j = ifac - 1
    return {"FOLD_EVEN": (n_taps // 2 - 1) + j * n_taps // 2, "FOLD_ODD": (n_taps - 1) // 2 + (n_taps // 2 + 1) * j,
            "FOLD_ANTI": (n_taps - 1) + n_taps * j}[ftype] + 1


def check(n_taps, ifac, ftype, fin, fc, fa, fo, n_ch=3, n=300, splits=None, seed=0, pairs=False):
    rng = np.random.default_rng(seed)
    csz = table_size(n_taps, ifac, ftype) + 2                       # COEFFSZ may exceed what the loops read
    c = rand_raw(rng, fc, (csz,))
    sign = rng.integers(0, 2, size=ifac)
    corr = np.arange(ifac)
    if pairs:                                                       # symmetric-pair technique: phase j paired with IF-1-j
        corr = ifac - 1 - corr
    x = rand_raw(rng, fin, (n_ch, n))
    eng = A.PolyIntr(n_taps, csz, ifac, ftype, fin, fc, fa, fo, n_channels=n_ch)
    eng.set_ctrl(c, sign, corr)
    orc = OraclePolyIntr(n_taps, csz, ifac, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    bounds = [0] + list(splits or []) + [n]
    for a, b in zip(bounds[:-1], bounds[1:]):
        xd = torch.from_numpy(x[:, a:b].copy()).to(A.torch_dtype_for(fin)).cuda()
        y = eng.run(xd).cpu().numpy().astype(np.int64)
        yo = orc.run(c, sign, corr, x[:, a:b])
        assert y.shape == yo.shape, (y.shape, yo.shape)
        bad = np.argwhere(y != yo)
        assert bad.size == 0, "%d mismatches, first at %s" % (len(bad), bad[0])
    return eng


@pytest.mark.parametrize("ftype,n_taps", [("FOLD_EVEN", 8), ("FOLD_EVEN", 7), ("FOLD_ODD", 9), ("FOLD_ODD", 6), ("FOLD_ANTI", 5)])
@pytest.mark.parametrize("ifac,pairs", [(1, False), (4, False), (4, True), (3, True)])
def test_usage_example_types_all_cores(ftype, n_taps, ifac, pairs):
    # types of the reference usage example (ac_poly_intr.h:44-48): <32,16> data / coefficients, <64,32> accumulator and output;
    # ragged bursts incl. a one-sample first burst (emits nothing on the folded cores)
    check(n_taps, ifac, ftype, A.Fmt(32, 16), A.Fmt(32, 16), A.Fmt(64, 32), A.Fmt(64, 32), splits=[1, 2, 50], seed=n_taps * 10 + ifac,
          pairs=pairs)


@pytest.mark.parametrize("ftype", ["FOLD_EVEN", "FOLD_ODD", "FOLD_ANTI"])
@pytest.mark.parametrize("q,o", [("TRN", "WRAP"), ("RND", "SAT"), ("RND_CONV", "SAT_SYM"), ("TRN_ZERO", "SAT_ZERO")])
def test_lossy_types_keep_every_quantisation_step(ftype, q, o):
    # narrow saturating ACC_TYPE: the IN_TYPE negation, the ACC_TYPE fold, every `acc +=`, the ACC_TYPE -t2 and the
    # final >> 1 all quantise; full-scale inputs so that -min(IN_TYPE) occurs
    fin, fc = A.Fmt(12, 3, True, q, o), A.Fmt(10, 2)
    fa, fo = A.Fmt(18, 7, True, q, o), A.Fmt(9, 5, True, q, o)
    eng = check(9 if ftype == "FOLD_ODD" else 8, 4, ftype, fin, fc, fa, fo, n=400, splits=[100], seed=5, pairs=True)
    # extreme stream: alternating full-scale values
    x = np.tile(np.array([-2048, 2047, -2048, -2048], dtype=np.int64), (1, 50))
    xd = torch.from_numpy(x).to(torch.int16).cuda()
    eng2 = A.PolyIntr(8, 40, 4, ftype, fin, fc, fa, fo, n_channels=1)
    rng = np.random.default_rng(9)
    c = rand_raw(rng, fc, (40,))
    sign, corr = [0, 1, 0, 1], [3, 2, 1, 0]
    eng2.set_ctrl(c, sign, corr)
    orc = OraclePolyIntr(8, 40, 4, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo))
    assert np.array_equal(eng2.run(xd).cpu().numpy().astype(np.int64), orc.run(c, sign, corr, x))


def test_control_reload_mid_stream_uses_the_sums_of_their_own_time():
    # new coefficients / control arrive between two samples: the group emitted by the next sample still carries the sums
    # formed with the OLD coefficients (acc banks, :153-161) but is combined with the NEW sign / corr
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12)
    rng = np.random.default_rng(3)
    x = rand_raw(rng, fin, (2, 64))
    eng = A.PolyIntr(8, 16, 4, "FOLD_EVEN", fin, fc, fa, fa, n_channels=2)
    orc = OraclePolyIntr(8, 16, 4, "FOLD_EVEN", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fa), n_ch=2)
    ys, yos = [], []
    for k, (a, b) in enumerate([(0, 20), (20, 21), (21, 64)]):
        c = rand_raw(np.random.default_rng(100 + k), fc, (16,))
        sign, corr = [k % 2, 1, 0, 1], ([3, 2, 1, 0] if k != 1 else [0, 1, 2, 3])
        eng.set_ctrl(c, sign, corr)
        ys.append(eng.run(torch.from_numpy(x[:, a:b].copy()).to(torch.int16).cuda()).cpu().numpy().astype(np.int64))
        yos.append(orc.run(c, sign, corr, x[:, a:b]))
    assert np.array_equal(np.concatenate(ys, axis=1), np.concatenate(yos, axis=1))


def test_rejects_tables_the_reference_would_overrun():
    fin, fa = A.Fmt(16, 2), A.Fmt(40, 12)
    eng = A.PolyIntr(8, 10, 4, "FOLD_EVEN", fin, fin, fa, fa)
    with pytest.raises(A.AcdspError):
        eng.set_ctrl(np.zeros(10, dtype=np.int64), [1, 1, 1, 1], [0, 1, 2, 3])      # needs coeffs[3 + 12] of coeffs[10]
    eng = A.PolyIntr(8, 16, 4, "FOLD_EVEN", fin, fin, fa, fa)
    with pytest.raises(A.AcdspError):
        eng.set_ctrl(np.zeros(16, dtype=np.int64), [1, 1, 1, 1], [0, 1, 2, 4])      # corr[3] = 4 outside the IF = 4 banks
    with pytest.raises(A.AcdspError):
        eng.run(torch.zeros((1, 16), dtype=torch.int16, device="cuda"))              # run before the control structs arrived


@pytest.mark.parametrize("ftype,n_taps", [("FOLD_EVEN", 16), ("FOLD_ODD", 15), ("FOLD_ANTI", 12)])
@pytest.mark.parametrize("in_o", ["WRAP", "SAT", "SAT_SYM", "SAT_ZERO"])
def test_lossless_class_int64_kernel(ftype, n_taps, in_o):
    # ACC <40,12> holds every product and the fold exactly (28 = 14 + 14 fraction bits): the exact-accumulation kernel.
    # Streams full of the most negative word exercise the one non-linear step left, the IN_TYPE negation (:145)
    fin = A.Fmt(16, 2, True, "TRN", in_o)
    fc, fa = A.Fmt(16, 2), A.Fmt(40, 12)
    for fo, pairs in ((A.Fmt(16, 2, True, "RND", "SAT"), True), (fa, False)):
        check(n_taps, 8, ftype, fin, fc, fa, fo, n_ch=4, n=500, splits=[1, 255], seed=n_taps, pairs=pairs)
    rng = np.random.default_rng(7)
    csz = table_size(n_taps, 4, ftype)
    c = rand_raw(rng, fc, (csz,))
    x = rng.choice(np.array([-32768, 32767, -32768, 0, 1, -1], dtype=np.int64), size=(2, 300))
    sign, corr = [0, 0, 1, 0], [1, 0, 3, 2]
    eng = A.PolyIntr(n_taps, csz, 4, ftype, fin, fc, fa, fa, n_channels=2)
    eng.set_ctrl(c, sign, corr)
    orc = OraclePolyIntr(n_taps, csz, 4, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fa), n_ch=2)
    y = eng.run(torch.from_numpy(x).to(torch.int16).cuda()).cpu().numpy().astype(np.int64)
    assert np.array_equal(y, orc.run(c, sign, corr, x))


# ---- matrix-core path (fir_up.hip): exact-accumulation class, int16 samples, all phases with sign set ----

def check_up(n_taps, ifac, ftype, fo, n_ch=3, n=16 * 33 * 3 + 80, splits=None, seed=0, pairs=True, coeff_bits=13, expect="mfma_gen", sign=None,
             fa=A.Fmt(40, 12), fin=A.Fmt(16, 2), fc=A.Fmt(16, 2)):
    rng = np.random.default_rng(seed)
    csz = table_size(n_taps, ifac, ftype) + 1
    c = rng.integers(-(1 << (coeff_bits - 1)), 1 << (coeff_bits - 1), size=csz, dtype=np.int64)
    sg = np.ones(ifac, dtype=np.int64) if sign is None else np.asarray(sign)
    corr = (ifac - 1 - np.arange(ifac)) if pairs else np.arange(ifac)
    x = rand_raw(rng, fin, (n_ch, n))
    x[0, :40] = -(1 << (fin.W - 1))                         # the most negative word everywhere in one window
    eng = A.PolyIntr(n_taps, csz, ifac, ftype, fin, fc, fa, fo, n_channels=n_ch)
    eng.set_ctrl(c, sg, corr)
    orc = OraclePolyIntr(n_taps, csz, ifac, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    bounds = [0] + list(splits or []) + [n]
    seen = set()
    for a, b in zip(bounds[:-1], bounds[1:]):
        xd = torch.from_numpy(x[:, a:b].copy()).to(A.torch_dtype_for(fin)).cuda()
        y = eng.run(xd).cpu().numpy().astype(np.int64)
        seen.add(eng.path)
        yo = orc.run(c, sg, corr, x[:, a:b])
        assert y.shape == yo.shape, (y.shape, yo.shape)
        bad = np.argwhere(y != yo)
        assert bad.size == 0, "%d mismatches, first at %s: got %d want %d" % (len(bad), bad[0], y[tuple(bad[0])], yo[tuple(bad[0])])
    if expect:
        assert expect in seen, seen
    return eng


@pytest.mark.parametrize("ftype,n_taps", [("FOLD_EVEN", 16), ("FOLD_ODD", 15), ("FOLD_ANTI", 16), ("FOLD_EVEN", 12), ("FOLD_ODD", 9), ("FOLD_EVEN", 40)])
@pytest.mark.parametrize("ifac", [2, 4, 8, 16])
def test_matrix_core_path_all_cores(ftype, n_taps, ifac):
    fo = A.Fmt(16, 2, True, "RND", "SAT")
    check_up(n_taps, ifac, ftype, fo, seed=n_taps + ifac, pairs=True)
    check_up(n_taps, ifac, ftype, fo, seed=n_taps + ifac + 1, pairs=False, splits=[8, 8 + 16 * 36, 8 + 16 * 72 + 3], n=16 * 33 * 3 + 160)   # state carry, ragged and unaligned bursts


@pytest.mark.parametrize("ftype,n_taps", [("FOLD_EVEN", 16), ("FOLD_ODD", 15), ("FOLD_ANTI", 16), ("FOLD_EVEN", 40)])
@pytest.mark.parametrize("ifac", [2, 4, 8, 16])
@pytest.mark.parametrize("coeff_bits", [15, 16])
def test_matrix_core_path_three_digit_planes(ftype, n_taps, ifac, coeff_bits):
    """Full-range coefficients: the folded pair taps need 17 bits = three digit planes (the 13-bit sets above stay inside two and
    run the two-plane instantiation)."""
    for fo in (A.Fmt(16, 2, True, "RND", "SAT"), A.Fmt(40, 12)):
        check_up(n_taps, ifac, ftype, fo, seed=100 + n_taps + ifac, coeff_bits=coeff_bits, splits=[8 + 16 * 36], n=16 * 33 * 3 + 160)


@pytest.mark.parametrize("fo", [A.Fmt(40, 12), A.Fmt(16, 2, True, "TRN", "WRAP"), A.Fmt(24, 6, True, "RND_CONV", "SAT_SYM"), A.Fmt(12, 4, False, "RND", "SAT"),
                                A.Fmt(24, 6, True, "RND", "SAT"), A.Fmt(32, 10, True, "TRN", "WRAP"), A.Fmt(36, 9, True, "RND", "SAT"), A.Fmt(48, 20, True, "RND", "WRAP"),
                                A.Fmt(50, 18, True, "TRN", "SAT"), A.Fmt(20, 3, True, "RND", "WRAP"), A.Fmt(33, 2, True, "TRN", "WRAP")])
def test_matrix_core_path_output_types(fo):
    # 2-, 4- and 8-byte output containers (4-byte ones since round 4, one K block only)
    check_up(16, 8, "FOLD_EVEN", fo, seed=5, n=16 * 33 * 5)
    check_up(15, 8, "FOLD_ODD", fo, seed=6, n=16 * 33 * 2 + 8, pairs=False)
    check_up(16, 2, "FOLD_EVEN", fo, seed=7, n=16 * 33 * 5, splits=[16 * 40])      # four steps per wave at IF = 2
    check_up(16, 4, "FOLD_ANTI", fo, seed=8, n=16 * 33 * 4, coeff_bits=16)         # three digit planes


def test_matrix_core_path_is_left_when_the_cores_are_not_linear():
    fo = A.Fmt(16, 2, True, "RND", "SAT")
    # a phase with sign = 0 negates samples in IN_TYPE (-min wraps): stays on the VALU kernel, results still exact
    check_up(16, 8, "FOLD_EVEN", fo, seed=7, sign=[1, 1, 0, 1, 1, 1, 1, 1], expect="lossless64")
    # an accumulator that can wrap (34 bits for sums of up to 2^35): the pair halving does not commute with the wrap
    check_up(16, 8, "FOLD_EVEN", fo, seed=8, coeff_bits=16, expect="lossless64", fa=A.Fmt(34, 6))
    check_up(16, 8, "FOLD_EVEN", fo, seed=8, coeff_bits=16, fa=A.Fmt(40, 12))
    # interpolation factors that are not compiled in fall back as well
    check_up(16, 9, "FOLD_EVEN", fo, seed=9, expect="lossless64")
    check_up(16, 32, "FOLD_EVEN", fo, seed=10, expect="lossless64")


@pytest.mark.parametrize("ftype,n_taps", [("FOLD_EVEN", 16), ("FOLD_ODD", 15), ("FOLD_ANTI", 12)])
@pytest.mark.parametrize("ifac", [2, 4, 8, 16, 3, 6, 7])
def test_matrix_core_path_on_the_usage_example_types(ftype, n_taps, ifac):
    """The header's own example (ac_poly_intr.h:44-48): <32,16> samples and coefficients, ACC = OUT = <64,32>.  Four input byte planes; a
    64-bit accumulator belongs to the exact-accumulation class when the control words bound every sub-filter sum to 62 bits.  Factors that
    do not divide 32 run SPC * IF live rows of the 32-row tile."""
    f32, a64 = A.Fmt(32, 16), A.Fmt(64, 32)
    n = 16 * 33 * 3 + 80
    check_up(n_taps, ifac, ftype, a64, seed=ifac + n_taps, coeff_bits=17, fa=a64, fin=f32, fc=f32, n=n, splits=[16 * 40])
    check_up(n_taps, ifac, ftype, A.Fmt(48, 20, True, "RND", "SAT"), seed=ifac, coeff_bits=16, fa=a64, fin=f32, fc=f32, pairs=False, n=n)
    # full-range 32-bit coefficients: sums past 62 bits -- the accumulator may wrap, exact-order kernel
    check_up(n_taps, ifac, ftype, a64, seed=ifac + 1, coeff_bits=32, fa=a64, fin=f32, fc=f32, n=700, expect="generic")
    # 4-byte containers on 4-byte samples are not compiled in
    check_up(n_taps, ifac, ftype, A.Fmt(32, 16, True, "RND", "SAT"), seed=ifac + 2, coeff_bits=16, fa=a64, fin=f32, fc=f32, n=n, expect="lossless64")


@pytest.mark.parametrize("ifac", [3, 5, 6, 7])
def test_matrix_core_path_factors_that_do_not_divide_32(ifac):
    for fo in (A.Fmt(16, 2, True, "RND", "SAT"), A.Fmt(40, 12), A.Fmt(24, 6, True, "RND", "SAT")):
        # (odd factors: the first call of a folded stream starts IF outputs early -- 2-byte runs then leave the 4-byte grid and stay on the VALU kernel;
        # a second call of whole steps is aligned again)
        check_up(16, ifac, "FOLD_EVEN", fo, seed=ifac, n=16 * 33 * 4 + 16, splits=[16 * 36], expect=None)
        check_up(12, ifac, "FOLD_ANTI", fo, seed=ifac + 1, n=16 * 33 * 3 + 8)
