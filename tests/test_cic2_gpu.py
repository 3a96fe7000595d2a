"""GPU parity tests of the two-stage CIC decimator (ac_dsp_amd/csrc/cic2.hip): ac_cic_dec_full at R = R1 R2 >= 32 -- the FIR identity of
rate R1 on the matrix cores, N prefix-sum integrators and the combs of rate R2 behind it in the same launch -- against the CPU oracle
(reference include/ac_dsp/ac_cic_full_core.h:80-135,228-255, ac_cic_dec_full.h:187-222), through the C ABI."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from oracle import OracleCic
from helpers import ofmt

pytestmark = pytest.mark.gpu


def rand_raw(rng, fmt, shape):
    lo = -(1 << (fmt.W - 1)) if fmt.S else 0
    hi = (1 << (fmt.W - 1)) - 1 if fmt.S else (1 << fmt.W) - 1
    return rng.integers(lo, hi + 1, size=shape, dtype=np.int64)


def run_calls(cic, x, lengths):
    """The stream x in calls of the given lengths, each from its own 16-byte aligned buffer whose row stride is the length rounded up to 16
    samples (what the matrix-core kernels ask of a row), so that calls of ANY length -- every decimation phase -- stay on them."""
    dt = A.torch_dtype_for(cic.fin)
    outs, paths, pos = [], [], 0
    for ln in lengths:
        seg = x[:, pos:pos + ln]
        pos += ln
        buf = torch.zeros((x.shape[0], (ln + 15) // 16 * 16 + 16), dtype=dt, device="cuda")
        buf[:, :ln] = torch.from_numpy(seg.copy()).to(dt).cuda()
        outs.append(cic.run(buf[:, :ln]).cpu().numpy().astype(np.int64))
        paths.append(cic.path)
    assert pos == x.shape[1]
    return np.concatenate(outs, axis=1), paths


def oracle_calls(R, M, N, fin, fout, x, lengths):
    orc = OracleCic(False, R, M, N, ofmt(fin), ofmt(fout), n_ch=x.shape[0])
    outs, pos = [], 0
    for ln in lengths:
        outs.append(orc.run(x[:, pos:pos + ln]))
        pos += ln
    return np.concatenate(outs, axis=1)


def int_fmt(R, M, N, fin):
    it = A.Cic(False, R, M, N, fin, fin).int_type
    return A.Fmt(it.W, it.I)


CASES = [
    # W, I, R, M, N      (stage-1 rates compiled: 16 / 10 / 8 / 15 / 5 / 12 / 6 / 7 / 4 / 3 on 16-bit samples, 10 / 8 / 5 / 4 / 3 / 6 / 7 on 32-bit samples: cic2.hip ACDSP_CIC2_SHAPES)
    (16, 1, 32, 1, 4), (16, 1, 64, 1, 3), (16, 1, 128, 2, 3), (16, 1, 256, 1, 3), (16, 1, 32, 1, 6), (16, 1, 48, 2, 3), (16, 1, 40, 1, 5),
    (16, 1, 64, 1, 5), (12, 4, 96, 1, 3), (16, 1, 100, 1, 3), (16, 1, 250, 1, 3), (16, 1, 255, 2, 3), (16, 1, 35, 1, 4), (16, 1, 45, 1, 5),
    (16, 1, 56, 1, 2), (16, 1, 200, 1, 1),
    (32, 16, 32, 1, 4), (32, 16, 64, 2, 3), (32, 16, 128, 1, 3), (32, 16, 72, 1, 3), (32, 16, 32, 2, 5), (24, 8, 64, 1, 4), (32, 16, 200, 1, 3),
    (32, 16, 100, 1, 4), (32, 16, 255, 2, 3), (32, 16, 250, 1, 3), (32, 16, 36, 1, 5), (32, 16, 33, 1, 4), (20, 3, 35, 2, 3),
    # second set of stage-1 rates (12 / 6 / 7 / 4 / 3 on 16-bit samples, 6 / 7 on 32-bit samples)
    (16, 1, 36, 1, 4), (16, 1, 42, 1, 3), (16, 1, 49, 2, 3), (16, 1, 44, 1, 4), (16, 1, 33, 1, 5), (16, 1, 252, 1, 3), (12, 4, 84, 2, 3), (16, 1, 63, 1, 4),
    (16, 1, 39, 2, 2), (32, 16, 42, 1, 4), (32, 16, 49, 1, 3), (32, 16, 66, 2, 3), (24, 8, 77, 1, 3),
    # rates below 32 whose R M N is too many taps for the one-stage FIR identity (they ran the recurrence kernel)
    (16, 1, 24, 2, 4), (32, 16, 24, 2, 5), (16, 1, 30, 1, 6), (32, 16, 27, 2, 3), (16, 1, 28, 2, 5), (16, 1, 14, 2, 6), (32, 16, 16, 2, 6), (16, 1, 25, 2, 3),
    (32, 16, 21, 2, 5), (16, 1, 15, 2, 6),
]


@pytest.mark.parametrize("W,I,R,M,N", CASES)
def test_two_stage_decimator_vs_oracle_ragged_calls(W, I, R, M, N):
    fin = A.Fmt(W, I)
    fout = int_fmt(R, M, N, fin)
    rng = np.random.default_rng(W * 1000 + R * 10 + N)
    # calls of odd lengths: the decimation phase of a call start runs through residues of every kind; a tiny call and an empty-output call between
    lengths = [70001, 3, 41003, R - 1, 36864 + 17, 5 * 4096, 1, 30011]
    n_ch = 3
    x = rand_raw(rng, fin, (n_ch, sum(lengths)))
    x[1, :20000] = (1 << (W - 1)) - 1        # full-scale runs: the integrators wrap many times over
    x[2, 5000:60000] = -(1 << (W - 1))
    cic = A.Cic(False, R, M, N, fin, fout, n_channels=n_ch)
    y, paths = run_calls(cic, x, lengths)
    yo = oracle_calls(R, M, N, fin, fout, x, lengths)
    assert y.shape == yo.shape
    assert np.array_equal(y, yo), np.argwhere(y != yo)[:5]
    assert paths[0] == "two_stage" and paths[2] == "two_stage", paths


@pytest.mark.parametrize("q,o,wo,io", [("RND", "SAT", 20, 4), ("TRN", "WRAP", 18, 2), ("RND_CONV", "SAT_SYM", 24, 12), ("TRN_ZERO", "SAT_ZERO", 30, 30)])
def test_two_stage_output_type_conversion(q, o, wo, io):
    fin, R, M, N = A.Fmt(16, 1), 64, 1, 4
    fout = A.Fmt(wo, io, True, q, o)
    rng = np.random.default_rng(wo)
    lengths = [50000, 33333]
    x = rand_raw(rng, fin, (2, sum(lengths)))
    cic = A.Cic(False, R, M, N, fin, fout, n_channels=2)
    y, paths = run_calls(cic, x, lengths)
    assert paths == ["two_stage", "two_stage"]
    assert np.array_equal(y, oracle_calls(R, M, N, fin, fout, x, lengths))


def test_two_stage_unsigned_samples_and_outputs():
    fin, R, M, N = A.Fmt(15, 15, False), 32, 2, 3
    it = A.Cic(False, R, M, N, fin, fin).int_type
    fout = A.Fmt(it.W - 1, it.I - 1, False)
    rng = np.random.default_rng(15)
    lengths = [40000, 40001]
    x = rand_raw(rng, fin, (2, sum(lengths)))
    cic = A.Cic(False, R, M, N, fin, fout, n_channels=2)
    y, paths = run_calls(cic, x, lengths)
    assert paths[0] == "two_stage"
    assert np.array_equal(y, oracle_calls(R, M, N, fin, fout, x, lengths))


def test_two_stage_and_recurrence_kernels_agree_and_share_state_blobs():
    """The same stream through a handle on the two-stage kernel and one held on the recurrence kernel (ACDSP_FLAG_FORCE_GENERIC), with
    the state blob of each loaded into the other half way: blobs of the two history lengths are interchangeable."""
    fin, R, M, N = A.Fmt(16, 1), 64, 2, 3
    fout = int_fmt(R, M, N, fin)
    rng = np.random.default_rng(7)
    lengths = [45001, 52003]
    x = rand_raw(rng, fin, (4, sum(lengths)))
    new = A.Cic(False, R, M, N, fin, fout, n_channels=4)
    old = A.Cic(False, R, M, N, fin, fout, n_channels=4, force_generic=True)
    y_new, p_new = run_calls(new, x, lengths)
    y_old, p_old = run_calls(old, x, lengths)
    assert p_new[0] == "two_stage" and p_old == ["recurrence", "recurrence"]
    assert np.array_equal(y_new, y_old)
    yo = oracle_calls(R, M, N, fin, fout, x, lengths)
    assert np.array_equal(y_new, yo)
    # first call on one handle, blob into the other kind of handle, second call there
    for a_kw, b_kw in (({}, {"force_generic": True}), ({"force_generic": True}, {})):
        a = A.Cic(False, R, M, N, fin, fout, n_channels=4, **a_kw)
        b = A.Cic(False, R, M, N, fin, fout, n_channels=4, **b_kw)
        y0, _ = run_calls(a, x[:, :lengths[0]], lengths[:1])
        blob = a.state()
        assert len(blob) != len(b.state())
        b.set_state(blob)
        y1, _ = run_calls(b, x[:, lengths[0]:], lengths[1:])
        assert np.array_equal(np.concatenate([y0, y1], axis=1), yo), (a_kw, b_kw)


def test_two_stage_many_channels_whole_chunks():
    """A bank wider than one wave row, rows of whole chunks (the bench geometry), distinct data per channel."""
    fin, R, M, N = A.Fmt(16, 1), 32, 1, 5
    fout = int_fmt(R, M, N, fin)
    rng = np.random.default_rng(99)
    n_ch, n = 130, 1 << 16
    x = rand_raw(rng, fin, (n_ch, n))
    cic = A.Cic(False, R, M, N, fin, fout, n_channels=n_ch)
    y, paths = run_calls(cic, x, [n])
    assert paths == ["two_stage"]
    assert np.array_equal(y, oracle_calls(R, M, N, fin, fout, x, [n]))


@pytest.mark.parametrize("W,I,R", [(16, 1, 37), (32, 16, 127), (16, 1, 34)])
def test_rates_no_compiled_stage_one_divides_stay_on_the_recurrence_kernel(W, I, R):
    """Primes (and 2 x 17 on 16-bit samples): acdsp_cic_path says so, and the outputs are the oracle's all the same."""
    fin, M, N = A.Fmt(W, I), 1, 3
    fout = int_fmt(R, M, N, fin)
    rng = np.random.default_rng(R)
    lengths = [40000, 20001]
    x = rand_raw(rng, fin, (2, sum(lengths)))
    cic = A.Cic(False, R, M, N, fin, fout, n_channels=2)
    y, paths = run_calls(cic, x, lengths)
    assert paths == ["recurrence", "recurrence"], paths
    assert np.array_equal(y, oracle_calls(R, M, N, fin, fout, x, lengths))
