"""BASELINE configs 3, 4 and 5 at the sizes bench.py runs them (per GPU): launches of many chunks per channel over
thousands of rows and multi-GiB offsets, which the reduced-size parity tests never issue.  The oracle cannot run these sizes
in seconds, so each is checked by
  (a) the oracle on sampled (channel, time) windows -- all three filters are FIR systems of their input (mod 2^W for the CIC),
      so an oracle started from zero state `warm` samples before a window reproduces the stream's outputs inside it, and
  (b) a size-independent property on a slice at full length: linearity y(x1 + x2) = y(x1) + y(x2) of the exact /
      wrap-arithmetic output."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from helpers import ofmt, windowed_sinc
from oracle import OracleCic, OracleFir, stimulus

pytestmark = pytest.mark.gpu
SEED = 0xACD5


def test_config3_full_size():
    """ac_cic_dec_full N5 R8 M1 <32,16> -> <47,31>, 4096 channels x 2^22 samples (68.7 GB in, 17 GB out)."""
    fin, fout = A.Fmt(32, 16), A.Fmt(47, 31)
    n_ch, n, R = 4096, 1 << 22, 8
    x = torch.empty((n_ch, n), dtype=torch.int32, device="cuda")
    A.fill_stimulus(x, SEED, 32)
    cic = A.Cic(False, R, 1, 5, fin, fout, n_channels=n_ch)
    y = cic.run(x)
    torch.cuda.synchronize()
    assert cic.path == "mfma_gen" and y.shape == (n_ch, n // R)
    warm, w = 64 * R, 512 * R                       # FIR memory of the identity: N (R M' - 1) + N = 40 inputs
    for ch, t0 in ((0, 0), (1, 8 * 1000), (2047, 8 * 262144 - 8 * 100), (4095, n - w), (3333, 8 * 400001)):
        ta = max(t0 - warm, 0)
        xs = stimulus(SEED, 1, t0 - ta + w, 32, ch0=ch, t0=ta)
        yo = OracleCic(False, R, 1, 5, ofmt(fin), ofmt(fout)).run(xs)[0][(t0 - ta) // R:]
        got = y[ch, t0 // R:(t0 + w) // R].cpu().numpy().astype(np.int64)
        assert np.array_equal(got, yo), (ch, t0)
    # linearity mod 2^47 on 16 rows at full length (halved inputs: the sum stays inside <32,16>)
    c16 = A.Cic(False, R, 1, 5, fin, fout, n_channels=16)
    x1, x2 = x[:16] >> 1, x[16:32] >> 1
    y1 = c16.run(x1).clone(); c16.reset()
    y2 = c16.run(x2).clone(); c16.reset()
    y12 = c16.run(x1 + x2)
    s = y1 + y2
    s = (s << 17) >> 17                              # wrap the int64 sum to 47 bits
    assert torch.equal(y12, s)


def test_config4_full_size_per_gpu():
    """ac_fir_prog_coeffs 1023 taps <16,2>, ACC <42,14>, 1024 channels x 2^20 samples (the per-GPU slice of config 4)."""
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(42, 14), A.Fmt(16, 2, True, "RND", "SAT")
    N, n_ch, n = 1023, 1024, 1 << 20
    c = windowed_sinc(N, 0.05, fc)
    x = torch.empty((n_ch, n), dtype=torch.int16, device="cuda")
    A.fill_stimulus(x, SEED, 16)
    fir = A.Fir(N, "SHIFT_REG", fin, fc, fa, fo, n_channels=n_ch, kind="prog")
    fir.set_coeffs(c)
    assert fir.path == "mfma_i8"
    y = fir.run(x)
    torch.cuda.synchronize()
    w = 2048
    for ch, t0 in ((0, 0), (3, 70000), (511, 524288 - 1000), (1023, n - w)):
        ta = max(t0 - (N - 1), 0)
        xs = stimulus(SEED, 1, t0 - ta + w, 16, ch0=ch, t0=ta)
        yo = OracleFir(N, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo)).run(c, xs)[0][t0 - ta:]
        assert np.array_equal(y[ch, t0:t0 + w].cpu().numpy().astype(np.int64), yo), (ch, t0)
    # second call: the 1024-sample history of every channel carries
    y2 = fir.run(x)
    xs = np.concatenate([stimulus(SEED, 1, n, 16, ch0=77)[:, n - (N - 1):], stimulus(SEED, 1, w, 16, ch0=77)], axis=1)
    yo = OracleFir(N, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo)).run(c, xs)[0][N - 1:]
    assert np.array_equal(y2[77, :w].cpu().numpy().astype(np.int64), yo)
    # linearity with the lossless output (OUT = ACC) on 32 rows at full length
    firw = A.Fir(N, "SHIFT_REG", fin, fc, fa, fa, n_channels=32, kind="prog")
    firw.set_coeffs(c)
    x1, x2 = x[:32] >> 1, x[32:64] >> 1
    y1 = firw.run(x1).clone(); firw.reset()
    y2w = firw.run(x2).clone(); firw.reset()
    assert torch.equal(firw.run(x1 + x2), y1 + y2w)


def test_config5_full_size_per_gpu():
    """DDC: ac_cic_dec_full R16 N5 on <16,1> -> 127-tap FIR on the 36-bit words, 4096 real streams x 2^20 samples."""
    cin, fc, fa, fo = A.Fmt(16, 1), A.Fmt(16, 1), A.Fmt(60, 30), A.Fmt(24, 9, True, "RND", "SAT")
    n_ch, n, R, NT = 4096, 1 << 20, 16, 127
    c = windowed_sinc(NT, 0.2, fc)
    x = torch.empty((n_ch, n), dtype=torch.int16, device="cuda")
    A.fill_stimulus(x, SEED, 16)
    ddc = A.Ddc(R, 1, 5, cin, NT, "SHIFT_REG", fc, fa, fo, n_channels=n_ch, kind="const")
    ddc.set_coeffs(c)
    assert ddc.path == "fused"
    y = ddc.run(x)
    torch.cuda.synchronize()
    assert y.shape[1] == n // R
    mid = ddc.int_type
    warm, w = (NT + 16) * R, 600 * R                # stage B needs 126 earlier stage-A outputs, stage A ~5 more
    for ch, t0 in ((0, 0), (5, 16 * 5000), (2048, 16 * 32768 - 16 * 300), (4095, n - w)):
        ta = max(t0 - warm, 0)
        xs = stimulus(SEED, 1, t0 - ta + w, 16, ch0=ch, t0=ta)
        u = OracleCic(False, R, 1, 5, ofmt(cin), ofmt(mid)).run(xs)
        yo = OracleFir(NT, "SHIFT_REG", ofmt(mid), ofmt(fc), ofmt(fa), ofmt(fo)).run(c, u)[0][(t0 - ta) // R:]
        got = y[ch, t0 // R:(t0 + w) // R].cpu().numpy().astype(np.int64)
        assert np.array_equal(got, yo), (ch, t0)


def test_mv_avg_bench_size():
    """ac_mv_avg TAPS 9 AC_MIRROR at the bench row's size (1024 objects x 1024 frames x 1024 samples): sampled frames against
    the oracle (frames are independent: no state crosses them), frame independence as a property (the same frame data placed at
    another (object, frame) position gives the same outputs), and linearity with the lossless OUT = ACC on full rows."""
    from oracle import OracleMvAvg
    fin, fc, fa, fo = A.Fmt(16, 8), A.Fmt(16, 2), A.Fmt(40, 18), A.Fmt(16, 8, True, "RND", "SAT")
    n_obj, n_frames, ns, taps = 1024, 1024, 1024, 9
    wts = np.round(np.hanning(11)[1:-1] / np.hanning(11).sum() * 2.0 ** fc.F).astype(np.int64)
    x = torch.empty((n_obj, n_frames * ns), dtype=torch.int16, device="cuda")
    A.fill_stimulus(x, SEED, 16)
    eng = A.MvAvg(ns, taps, "MIRROR", fin, fc, fa, fo, n_objects=n_obj)
    eng.set_coeffs(wts)
    y = eng.run(x, ns)
    torch.cuda.synchronize()
    assert eng.path == "stream" and y.shape == x.shape
    orc = OracleMvAvg(taps, "MIRROR", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_obj=1)
    for obj, fr in ((0, 0), (0, 1023), (1, 1), (511, 512), (1023, 0), (1023, 1023), (700, 333)):
        xs = x[obj:obj + 1, fr * ns:(fr + 1) * ns].cpu().numpy().astype(np.int64)
        assert np.array_equal(y[obj, fr * ns:(fr + 1) * ns].cpu().numpy().astype(np.int64), orc.run(wts, xs, ns)[0]), (obj, fr)
    # frame independence: rows rolled by whole frames give rolled outputs
    e16 = A.MvAvg(ns, taps, "MIRROR", fin, fc, fa, fo, n_objects=16)
    e16.set_coeffs(wts)
    xr = torch.roll(x[:16], shifts=5 * ns, dims=1).contiguous()
    assert torch.equal(e16.run(xr, ns), torch.roll(y[:16], shifts=5 * ns, dims=1))
    # linearity of the exact accumulator (OUT = ACC <40,18>: no rounding, no saturation)
    ew = A.MvAvg(ns, taps, "MIRROR", fin, fc, fa, fa, n_objects=16)
    ew.set_coeffs(wts)
    x1, x2 = x[:16] >> 1, x[16:32] >> 1
    assert torch.equal(ew.run(x1 + x2, ns), ew.run(x1, ns) + ew.run(x2, ns))
