"""GPU parity tests, DDC cascade: acdsp_ddc_* (fused CIC-decimator -> FIR kernel, and its two-kernel
fallback) vs the oracle cascade OracleCic -> OracleFir on the same streams."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from oracle import OracleCic, OracleFir
from helpers import ofmt
from test_fir_gpu import windowed_sinc

pytestmark = pytest.mark.gpu


def oracle_cascade(R, M, N, cin, mid, n_taps, ftype, fc, fa, fo, c, x, splits=None):
    n_ch = x.shape[0]
    cic = OracleCic(False, R, M, N, ofmt(cin), ofmt(mid), n_ch=n_ch)
    fir = OracleFir(n_taps, ftype, ofmt(mid), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    outs = []
    bounds = [0] + list(splits or []) + [x.shape[1]]
    for a, b in zip(bounds[:-1], bounds[1:]):
        u = cic.run(x[:, a:b])
        outs.append(fir.run(c, u) if u.shape[1] else np.zeros((n_ch, 0), dtype=np.int64))
    return np.concatenate(outs, axis=1)


def run_ddc(ddc, x, splits=None):
    outs = []
    bounds = [0] + list(splits or []) + [x.shape[1]]
    for a, b in zip(bounds[:-1], bounds[1:]):
        xd = torch.from_numpy(x[:, a:b].copy()).to(A.torch_dtype_for(ddc.fin)).cuda()
        outs.append(ddc.run(xd).cpu().numpy().astype(np.int64))
    return np.concatenate(outs, axis=1)


CFG5 = dict(R=16, M=1, N=5, cin=A.Fmt(16, 1), n_taps=127, fc=A.Fmt(16, 1), fa=A.Fmt(60, 30))


# 32-bit limb epilogues (OUT <= 31 bits, shift <= 31: AC_SAT with the clamp in the high limb, AC_WRAP) and the 64-bit form (32-bit OUT)
@pytest.mark.parametrize("fo", [A.Fmt(24, 9, True, "RND", "SAT"), A.Fmt(24, 9, True, "TRN", "WRAP"), A.Fmt(32, 12, True, "RND", "WRAP"),
                                A.Fmt(20, 3, True, "TRN", "SAT"), A.Fmt(31, 12, True, "RND", "SAT"), A.Fmt(18, 17, True, "RND", "SAT"),
                                A.Fmt(30, 29, True, "RND", "WRAP")])
def test_config5_fused_cascade_matches_the_oracle_cascade(fo):
    g = CFG5
    rng = np.random.default_rng(17)
    n_ch, n = 3, 16 * (256 * 9 + 77)           # complete chunks of 256-output steps, a ragged last chunk
    x = rng.integers(-32768, 32768, size=(n_ch, n))
    c = windowed_sinc(g["n_taps"], 0.2, g["fc"])
    ddc = A.Ddc(g["R"], g["M"], g["N"], g["cin"], g["n_taps"], "SHIFT_REG", g["fc"], g["fa"], fo, n_channels=n_ch)
    assert (ddc.int_type.W, ddc.int_type.I) == (36, 21)
    ddc.set_coeffs(c)
    assert ddc.path == "fused"
    y = run_ddc(ddc, x)
    yo = oracle_cascade(g["R"], g["M"], g["N"], g["cin"], ddc.int_type, g["n_taps"], "SHIFT_REG", g["fc"], g["fa"], fo, c, x)
    assert y.shape == yo.shape and np.array_equal(y, yo)


def test_fused_cascade_state_carries_across_calls_and_decimation_phases():
    g = CFG5
    fo = A.Fmt(24, 9, True, "RND", "SAT")
    rng = np.random.default_rng(18)
    n_ch = 2
    # bursts of whole 16-sample slots but not of whole decimation-by-16 x 16 periods... plus one burst that leaves the
    # decimation phase mid-period (16 * k + 32 keeps 16-sample alignment; phase stays 0), and short bursts below one step
    splits = [16 * 300, 16 * 300 + 4096 * 5, 16 * 300 + 4096 * 5 + 48]
    n = splits[-1] + 16 * 2000
    x = rng.integers(-32768, 32768, size=(n_ch, n))
    c = windowed_sinc(g["n_taps"], 0.2, g["fc"])
    ddc = A.Ddc(g["R"], g["M"], g["N"], g["cin"], g["n_taps"], "SHIFT_REG", g["fc"], g["fa"], fo, n_channels=n_ch)
    ddc.set_coeffs(c)
    y = run_ddc(ddc, x, splits)
    yo = oracle_cascade(g["R"], g["M"], g["N"], g["cin"], ddc.int_type, g["n_taps"], "SHIFT_REG", g["fc"], g["fa"], fo, c, x, splits)
    assert np.array_equal(y, yo)
    ddc.reset()
    assert np.array_equal(run_ddc(ddc, x[:, :16 * 1024]), yo[:, :1024])


def test_other_shapes_take_the_two_kernels_and_match(monkeypatch):
    # 8-bit narrower decimator (R = 8 on <32,16>): int32 input containers -> not the fused shape class
    cin, fc, fa, fo = A.Fmt(32, 16), A.Fmt(16, 1), A.Fmt(64, 31), A.Fmt(32, 16, True, "RND", "SAT")
    rng = np.random.default_rng(19)
    x = rng.integers(-(1 << 31), 1 << 31, size=(2, 8 * 3000))
    c = windowed_sinc(63, 0.2, fc)
    ddc = A.Ddc(8, 1, 5, cin, 63, "SHIFT_REG", fc, fa, fo, n_channels=2)
    ddc.set_coeffs(c)
    assert ddc.path == "two_kernels"
    y = run_ddc(ddc, x, [8 * 1000])
    yo = oracle_cascade(8, 1, 5, cin, ddc.int_type, 63, "SHIFT_REG", fc, fa, fo, c, x, [8 * 1000])
    assert np.array_equal(y, yo)
    # the config-5 shape forced onto the two kernels gives the same stream as the fused kernel
    monkeypatch.setenv("ACDSP_NO_FUSE", "1")
    g = CFG5
    fo5 = A.Fmt(24, 9, True, "RND", "SAT")
    x5 = rng.integers(-32768, 32768, size=(2, 16 * 3000))
    c5 = windowed_sinc(127, 0.2, g["fc"])
    d2 = A.Ddc(16, 1, 5, g["cin"], 127, "SHIFT_REG", g["fc"], g["fa"], fo5, n_channels=2)
    d2.set_coeffs(c5)
    yo5 = oracle_cascade(16, 1, 5, g["cin"], d2.int_type, 127, "SHIFT_REG", g["fc"], g["fa"], fo5, c5, x5)
    assert np.array_equal(run_ddc(d2, x5), yo5)


def test_rejects_a_lossy_middle():
    with pytest.raises(A.AcdspError):
        cd = A.CicDesc(0, 16, 1, 5, 1, A.Fmt(16, 1), A.Fmt(30, 15), 0, 0)      # decimator output narrower than its INT_TYPE
        fd = A.FirDesc(0, 0, 127, 1, 0, A.Fmt(30, 15), A.Fmt(16, 1), A.Fmt(60, 30), A.Fmt(24, 9), 0, 0)
        h = A.lib.acdsp_ddc_create
        import ctypes as C
        out = C.c_void_p()
        from ac_dsp_amd._lib import check
        check(h(C.byref(cd), C.byref(fd), C.byref(out)))


def test_state_round_trip_fused_and_two_kernels():
    """acdsp_ddc_state_get / _set: run -> get -> fresh handle -> set -> continue == the uninterrupted stream == the oracle cascade."""
    rng = np.random.default_rng(23)
    cases = [
        (CFG5["R"], CFG5["cin"], 127, CFG5["fc"], CFG5["fa"], A.Fmt(24, 9, True, "RND", "SAT"), "fused", 16 * 700, 16 * 900),
        (8, A.Fmt(32, 16), 63, A.Fmt(16, 1), A.Fmt(64, 31), A.Fmt(32, 16, True, "RND", "SAT"), "two_kernels", 8 * 1000 + 3, 8 * 1200),
    ]
    for R, cin, n_taps, fc, fa, fo, path, n1, n2 in cases:
        lo, hi = -(1 << (cin.W - 1)), 1 << (cin.W - 1)
        x = rng.integers(lo, hi, size=(3, n1 + n2))
        c = windowed_sinc(n_taps, 0.2, fc)
        a = A.Ddc(R, 1, 5, cin, n_taps, "SHIFT_REG", fc, fa, fo, n_channels=3)
        a.set_coeffs(c)
        assert a.path == path
        yo = oracle_cascade(R, 1, 5, cin, a.int_type, n_taps, "SHIFT_REG", fc, fa, fo, c, x, [n1])
        y1 = run_ddc(a, x[:, :n1])
        blob = a.state()
        assert blob[:8] == b"ACDSPST1" and len(blob) == A.lib.acdsp_ddc_state_size(a._h)
        b = A.Ddc(R, 1, 5, cin, n_taps, "SHIFT_REG", fc, fa, fo, n_channels=3)
        b.set_coeffs(c)
        b.set_state(blob)
        y2 = run_ddc(b, x[:, n1:])
        assert np.array_equal(np.concatenate([y1, y2], axis=1), yo)
        assert np.array_equal(run_ddc(a, x[:, n1:]), y2)
        other = A.Ddc(R, 1, 5, cin, n_taps, "SHIFT_REG", fc, fa, fo, n_channels=2)
        other.set_coeffs(c)
        with pytest.raises(A.AcdspError):
            other.set_state(blob)
        with pytest.raises(A.AcdspError):
            b.set_state(blob[:-1])
