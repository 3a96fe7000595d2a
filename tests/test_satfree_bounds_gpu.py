"""Saturating accumulators exactly AT the bound that decides whether they can saturate (advisor, round 5).  A handle whose worst-case partial
sum stays inside the accumulator's range runs the wrapping-accumulator kernels (matrix cores / streaming kernels: the saturation is dead code),
one LSB more and it keeps the exact-order saturating kernels; an off-by-one in those hand-derived 128-bit bounds would turn a saturating
result into a wrapped one without any other test noticing.  For ac_poly_dec, ac_intg_dump and ac_mv_avg (the FIR classes:
tests/test_fir_gpu.py::test_saturating_accumulators_that_cannot_saturate_run_the_wrapping_classes): coefficient sets / block lengths with
sum|c| max|x| exactly at top, at top + 1, and far enough beyond that the full-scale input really saturates -- full-scale inputs of both
signs, every result against the oracle, and the path each handle took."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from helpers import ofmt
from oracle import OracleIntgDump, OracleMvAvg, OraclePolyDec

pytestmark = pytest.mark.gpu


def full_scale(fmt, shape, rng):
    """rows: all max, all min, alternating, random"""
    lo, hi = (-(1 << (fmt.W - 1)), (1 << (fmt.W - 1)) - 1) if fmt.S else (0, (1 << fmt.W) - 1)
    x = rng.integers(lo, hi + 1, size=shape, dtype=np.int64)
    x[0] = hi
    x[1] = lo
    x[2, ::2] = hi
    x[2, 1::2] = lo
    return x


@pytest.mark.parametrize("sum_abs,expect", [(65535, "mfma_gen"), (65536, "generic"), (65537, "generic"), (90000, "generic")])
@pytest.mark.parametrize("o", ["SAT", "SAT_SYM", "SAT_ZERO"])
def test_poly_dec_at_the_bound(sum_abs, expect, o):
    # <16,2> x <16,2> into ACC <32,4>: F_acc = F_in + F_c, top = 2^31 - 1, max|x| = 2^15: inside the range iff sum|c| < 2^16
    nt, df = 8, 4
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(32, 4, True, "TRN", o), A.Fmt(32, 4)
    c = np.zeros(nt * df, dtype=np.int64)
    rest = sum_abs
    for i in range(len(c)):                      # all positive: a full-scale row drives the sum straight to the bound
        c[i] = min(rest, 32767)
        rest -= c[i]
    assert rest == 0 and np.abs(c).sum() == sum_abs
    rng = np.random.default_rng(sum_abs)
    x = full_scale(fin, (4, 64 * df * 4), rng)
    eng = A.PolyDec(nt, df, fin, fc, fa, fo, n_channels=4)
    eng.set_coeffs(c)
    y = eng.run(torch.from_numpy(x).to(torch.int16).cuda()).cpu().numpy().astype(np.int64)
    assert eng.path == expect, (eng.path, expect)
    yo = OraclePolyDec(nt, df, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=4).run(c, x)
    assert np.array_equal(y, yo)
    if sum_abs >= 65537 and o == "SAT":
        assert (yo[1] == -(1 << 31)).any()       # the all-min row really hits the rail


@pytest.mark.parametrize("rounds,wacc,expect", [(64, 23, "wrapping"), (63, 22, "wrapping"), (64, 22, "exact_order"), (65, 22, "exact_order"), (127, 23, "wrapping"), (128, 23, "exact_order"), (129, 23, "exact_order")])
def test_intg_dump_at_the_bound(rounds, wacc, expect):
    # sums of `rounds` samples of <16,8> into ACC <wacc, wacc-8, SAT>: rounds * 2^15 <= 2^(wacc-1) - 1 ?
    ns, chn, n_obj = 256, 4, 4
    fin, fa, fo = A.Fmt(16, 8), A.Fmt(wacc, wacc - 8, True, "TRN", "SAT"), A.Fmt(32, 24)
    n_sample = [rounds] * 64
    rng = np.random.default_rng(rounds * 100 + wacc)
    x = full_scale(fin, (n_obj, rounds * chn * len(n_sample)), rng)
    eng = A.IntgDump(ns, chn, fin, fa, fo, n_objects=n_obj)
    y = eng.run(torch.from_numpy(x).to(torch.int16).cuda(), n_sample).cpu().numpy().astype(np.int64)
    assert (eng.path in ("stream", "tile")) if expect == "wrapping" else (eng.path == expect), (eng.path, expect)   # stream / tile: the kernels of a wrapping accumulator
    yo = OracleIntgDump(ns, chn, ofmt(fin), ofmt(fa), ofmt(fo), n_obj=n_obj).run(x, n_sample)
    assert np.array_equal(y, yo)
    if (rounds, wacc) in ((65, 22), (129, 23)):
        assert (yo[1] == -(1 << (wacc - 1))).all()   # 65 * -2^15 < -2^21: every sum of the all-min row saturates


@pytest.mark.parametrize("sum_abs,expect", [(262138, "wrapping"), (262139, "exact_order"), (262143, "exact_order")])
def test_mv_avg_at_the_bound(sum_abs, expect):
    # <16,8> samples, <16,2> weights, ACC <20,12,TRN,SAT> (F_acc = F_in): bound 2 sum|c| + TAPS + 1 <= 2^19 - 1 (engine_misc.hip)
    taps = 9
    fin, fc, fa, fo = A.Fmt(16, 8), A.Fmt(16, 2), A.Fmt(20, 12, True, "TRN", "SAT"), A.Fmt(20, 12)
    c = np.zeros(taps, dtype=np.int64)
    rest = sum_abs
    for i in range(taps):
        c[i] = min(rest, 32767)
        rest -= c[i]
    assert rest == 0
    rng = np.random.default_rng(sum_abs)
    n_sample, n_frames = 1024, 4
    x = full_scale(fin, (4, n_sample * n_frames), rng)
    eng = A.MvAvg(4096, taps, "MIRROR", fin, fc, fa, fo, n_objects=4)
    eng.set_coeffs(c)
    y = eng.run(torch.from_numpy(x).to(torch.int16).cuda(), n_sample).cpu().numpy().astype(np.int64)
    assert (eng.path in ("stream", "int64_sums", "stream32")) if expect == "wrapping" else (eng.path == expect), (eng.path, expect)
    yo = OracleMvAvg(taps, "MIRROR", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_obj=4).run(c, x, n_sample)
    assert np.array_equal(y, yo)
