"""GPU parity tests, polyphase decimator: HIP engine vs the CPU oracle restatement of
ac_poly_dec::run (reference include/ac_dsp/ac_poly_dec.h:109-128).  The reference ships no vectors for this
block, so parity is oracle-only ("parity unpinned" by reference data)."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from oracle import OraclePolyDec, stimulus
from helpers import ofmt

pytestmark = pytest.mark.gpu


def rand_raw(rng, fmt, shape):
    lo = -(1 << (fmt.W - 1)) if fmt.S else 0
    hi = (1 << (fmt.W - 1)) - 1 if fmt.S else (1 << fmt.W) - 1
    return rng.integers(lo, hi + 1, size=shape, dtype=np.int64)


def check(nt, df, fin, fc, fa, fo, n_ch=3, n_groups=500, splits=None, expect=None, seed=0):
    rng = np.random.default_rng(seed)
    c = rand_raw(rng, fc, (nt * df,))
    x = rand_raw(rng, fin, (n_ch, n_groups * df))
    eng = A.PolyDec(nt, df, fin, fc, fa, fo, n_channels=n_ch)
    eng.set_coeffs(c)
    orc = OraclePolyDec(nt, df, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    bounds = [0] + [s * df for s in (splits or [])] + [x.shape[1]]
    for a, b in zip(bounds[:-1], bounds[1:]):
        xd = torch.from_numpy(x[:, a:b].copy()).to(A.torch_dtype_for(fin)).cuda()
        y = eng.run(xd).cpu().numpy().astype(np.int64)
        if expect:
            assert eng.path == expect, eng.path
        assert np.array_equal(y, orc.run(c, x[:, a:b]))


@pytest.mark.parametrize("nt,df", [(8, 4), (5, 3), (16, 2), (3, 8), (7, 1), (4, 16)])
def test_lossless_class_runs_on_the_matrix_cores(nt, df):
    # usage-example types of the reference header (ac_poly_dec.h:45-48): <32,16> data/coefficients, <64,32> accumulator
    check(nt, df, A.Fmt(32, 16), A.Fmt(32, 16), A.Fmt(64, 32), A.Fmt(64, 32), splits=[1, 100], seed=nt * 10 + df)
    # burst lengths that are multiples of 16 samples keep the rows slot-aligned -> matrix-core kernel
    check(nt, df, A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(44, 16), A.Fmt(16, 2, True, "RND", "SAT"), n_groups=704, expect="mfma_gen",
          splits=[32], seed=nt + df)
    # ragged bursts (rows not 16-byte aligned) take the exact VALU kernel and must give the same stream
    check(nt, df, A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(44, 16), A.Fmt(16, 2, True, "RND", "SAT"), n_groups=300, splits=[33, 34], seed=100 + nt - df)


@pytest.mark.parametrize("q,o", [("TRN", "WRAP"), ("RND_CONV", "SAT"), ("TRN_ZERO", "SAT_SYM"), ("RND_INF", "SAT_ZERO")])
def test_lossy_accumulator_keeps_the_reference_mac_order(q, o):
    check(6, 4, A.Fmt(14, 4), A.Fmt(12, 2), A.Fmt(20, 8, True, q, o), A.Fmt(10, 5, True, q, o), splits=[7], expect="generic", seed=3)


def test_rejects_partial_groups():
    eng = A.PolyDec(4, 4, A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2))
    eng.set_coeffs(np.ones(16, dtype=np.int64))
    x = torch.zeros((1, 10), dtype=torch.int16, device="cuda")
    import ctypes as C
    rc = A.lib.acdsp_polydec_run(eng._h, C.c_void_p(x.data_ptr()), 10, 10, C.c_void_p(x.data_ptr()), 10, None)
    assert rc == 1   # ACDSP_EINVAL: run() only ever consumes whole groups of DF samples


def test_long_burst_takes_the_branch_free_kernel():
    # the bench row: 16 taps per branch x DF 8 on ac_fixed<16,2>, complete 256-output steps + ragged tail
    check(16, 8, A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(44, 16), A.Fmt(16, 2, True, "RND", "SAT"), n_groups=4096 + 70, expect="mfma_gen",
          splits=[2048], seed=5)
    check(16, 8, A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(44, 16), A.Fmt(16, 4, True, "TRN", "WRAP"), n_groups=3000, expect="mfma_gen", seed=6)


@pytest.mark.parametrize("nt,df", [(16, 2), (8, 2), (16, 4), (8, 4), (16, 8), (16, 16), (8, 16)])
@pytest.mark.parametrize("fo", [A.Fmt(16, 2, True, "RND", "SAT"), A.Fmt(44, 16), A.Fmt(16, 3, True, "TRN", "WRAP")])
def test_ring_kernel_shapes_outside_the_bench_row(nt, df, fo):
    """Decimation factors 2 / 4 / 8 / 16 into 2- and 8-byte containers: whole chunks of 2 - 8 steps on fir_gen_ring_kernel, the ragged
    tail and the second call's head (history in front of its first window) on the general kernel; every output against the oracle."""
    groups = 256 * 8 * 3 + 200          # three chunks of the longest shape (eight 256-output steps) + a ragged tail
    check(nt, df, A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(44, 16), fo, n_groups=groups, expect="mfma_gen", splits=[256 * 9 + 16], seed=nt * df + fo.W)


@pytest.mark.parametrize("o", ["SAT", "SAT_SYM", "SAT_ZERO"])
def test_saturating_accumulator_that_cannot_saturate_runs_on_the_matrix_cores(o):
    """A signed AC_SAT* ACC_TYPE whose bounds sum|c| * max|x| cannot reach is a wrapping one (acdsp_polydec_set_coeffs decides per set); a set that
    could reach them keeps the exact-order kernel -- and does saturate with full-scale inputs."""
    fin, fc, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(16, 2, True, "RND", "SAT")
    check(16, 8, fin, fc, A.Fmt(60, 32, True, "TRN", o), fo, n_groups=704, expect="mfma_gen", splits=[32], seed=5)     # 128 * 2^15 * 2^15 << 2^59
    check(16, 8, fin, fc, A.Fmt(34, 6, True, "TRN", o), fo, n_groups=704, expect="generic", splits=[32], seed=6)       # 2^37 > 2^33: exact order
