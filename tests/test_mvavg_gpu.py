"""GPU parity tests, ac_mv_avg: the HIP engine through the C ABI against the CPU oracle, bit for bit."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from helpers import ofmt
from oracle import OracleMvAvg

pytestmark = pytest.mark.gpu


def rand_raw(rng, fmt, shape):
    lo = -(1 << (fmt.W - 1)) if fmt.S else 0
    hi = (1 << (fmt.W - 1)) - 1 if fmt.S else (1 << fmt.W) - 1
    return rng.integers(lo, hi + 1, size=shape, dtype=np.int64)


def check(taps, mode, fin, fc, fa, fo, n_sample, n_frames, n_obj=3, seed=0, force_generic=False, max_sample=None):
    rng = np.random.default_rng(seed)
    x = rand_raw(rng, fin, (n_obj, n_sample * n_frames))
    c = rand_raw(rng, fc, (taps,))
    eng = A.MvAvg(max_sample or max(n_sample, 1), taps, mode, fin, fc, fa, fo, n_objects=n_obj, force_generic=force_generic)
    eng.set_coeffs(c)
    y = eng.run(torch.from_numpy(x).to(A.torch_dtype_for(fin)).cuda(), n_sample).cpu().numpy().astype(np.int64)
    yo = OracleMvAvg(taps, mode, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_obj=n_obj).run(c, x, n_sample)
    assert y.shape == yo.shape, (y.shape, yo.shape)
    bad = np.argwhere(y != yo)
    assert bad.size == 0, "%d mismatches, first at %s" % (len(bad), bad[0])


MODES = ["WIN", "MIRROR", "CLIP"]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("force_generic", [False, True])
def test_wrapping_accumulator_types(mode, force_generic):
    fin, fc, fa, fo = A.Fmt(16, 8), A.Fmt(16, 1), A.Fmt(32, 14), A.Fmt(20, 10, True, "RND", "SAT")
    check(9, mode, fin, fc, fa, fo, 1000, 3, force_generic=force_generic, seed=1)
    check(65, mode, fin, fc, A.Fmt(32, 14, True, "RND"), fa, 700, 2, force_generic=force_generic, seed=2)   # tiles narrower than the window reach
    check(1, mode, fin, fc, fa, fo, 300, 2, force_generic=force_generic, seed=3)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("q,o", [("RND_CONV", "SAT"), ("TRN_ZERO", "SAT_SYM"), ("RND_INF", "WRAP"), ("TRN", "SAT_ZERO")])
def test_order_dependent_accumulators(mode, q, o):
    check(7, mode, A.Fmt(24, 12), A.Fmt(12, 2), A.Fmt(16, 6, True, q, o), A.Fmt(14, 5, True, q, o), 257, 3, seed=4)   # the cast to ACC loses bits
    check(5, mode, A.Fmt(12, 4, False), A.Fmt(10, 0, False), A.Fmt(24, 10, False, q, o), A.Fmt(24, 10, False), 90, 4, seed=5)


@pytest.mark.parametrize("mode", MODES)
def test_short_frames_and_many_frames(mode):
    fin, fc, fa, fo = A.Fmt(16, 8), A.Fmt(16, 1), A.Fmt(32, 14), A.Fmt(32, 14)
    for n in (1, 2, 3, 4, 5, 8, 9, 10):
        check(9, mode, fin, fc, fa, fo, n, 11, seed=10 + n, max_sample=64)
    check(5, mode, fin, fc, fa, fo, 7, 3000, n_obj=2, seed=30, max_sample=64)     # more (object, frame) pairs than the grid has rows
    check(33, mode, fin, fc, fa, fo, 256, 5, seed=31)                              # frame = one tile exactly
    check(33, mode, fin, fc, fa, fo, 257, 5, seed=32)


def test_argument_checks():
    fin, fc, fa, fo = A.Fmt(16, 8), A.Fmt(16, 1), A.Fmt(32, 14), A.Fmt(32, 14)
    with pytest.raises(A.AcdspError):
        A.MvAvg(64, 8, "WIN", fin, fc, fa, fo)                # even TAPS
    eng = A.MvAvg(64, 5, "CLIP", fin, fc, fa, fo)
    x = torch.zeros((1, 130), dtype=torch.int16, device="cuda")
    with pytest.raises(A.AcdspError):
        eng.run(x, 65)                                        # before set_coeffs
    eng.set_coeffs(np.arange(5))
    with pytest.raises(A.AcdspError):
        eng.run(x, 65)                                        # n_sample > MAX_SAMPLE
    assert eng.run(x[:, :128], 64).shape == (1, 128)
    assert eng.out_per_frame(0) == -1 and eng.out_per_frame(64) == 64
