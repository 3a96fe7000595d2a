"""State save / restore through the C ABI (acdsp_{fir,cic}_state_get/set): run -> get -> new handle -> set -> continue must
equal the uninterrupted run, and equal the oracle (whose object state carries across run() calls like the reference's)."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from helpers import ofmt
from oracle import OracleFir, OracleCic

pytestmark = pytest.mark.gpu


def rand_raw(rng, fmt, shape):
    lo = -(1 << (fmt.W - 1)) if fmt.S else 0
    hi = (1 << (fmt.W - 1)) - 1 if fmt.S else (1 << fmt.W) - 1
    return rng.integers(lo, hi + 1, size=shape, dtype=np.int64)


def dev(x, fmt):
    return torch.from_numpy(np.ascontiguousarray(x)).to(A.torch_dtype_for(fmt)).cuda()


@pytest.mark.parametrize("ftype,kind,n_taps", [("SHIFT_REG", "load", 255), ("FOLD_ODD", "const", 63), ("TRANSPOSED", "load", 40),
                                               ("C_BUFF", "prog", 17), ("TRANSPOSED", "const", 33),
                                               ("SHIFT_REG", "load", 258), ("SHIFT_REG", "prog", 352)])   # padded plans: longer history
def test_fir_state_round_trip(ftype, kind, n_taps):
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
    rng = np.random.default_rng(3)
    n_ch, n1, n2 = 5, 1500, 1300
    x = rand_raw(rng, fin, (n_ch, n1 + n2))
    c = rand_raw(rng, fc, (n_taps,)) // 4
    want = OracleFir(n_taps, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch).run(c, x)

    a = A.Fir(n_taps, ftype, fin, fc, fa, fo, n_channels=n_ch, kind=kind)
    a.set_coeffs(c)
    y1 = a.run(dev(x[:, :n1], fin)).cpu().numpy().astype(np.int64)
    blob = a.state()
    assert blob[:8] == b"ACDSPST1" and len(blob) == A.lib.acdsp_fir_state_size(a._h)
    b = A.Fir(n_taps, ftype, fin, fc, fa, fo, n_channels=n_ch, kind=kind)
    b.set_coeffs(c)
    b.set_state(blob)
    y2 = b.run(dev(x[:, n1:], fin)).cpu().numpy().astype(np.int64)
    y2_same = a.run(dev(x[:, n1:], fin)).cpu().numpy().astype(np.int64)
    assert np.array_equal(y2, y2_same), "restored handle diverges from the uninterrupted one"
    assert np.array_equal(np.concatenate([y1, y2], axis=1), want)
    assert b.state() == a.state()


@pytest.mark.parametrize("n_taps,delta", [(240, 32), (240, -16), (353, 32), (63, 96)])
def test_fir_state_blob_of_another_history_length_restores(n_taps, delta):
    """The history length is n_taps - 1 rounded up for the kernels' windows and has changed between builds (round 3 kept 288 samples
    for 240 taps, round 4 keeps 256).  A blob with a longer or shorter history that still covers the taps is this filter's state:
    the newest samples are kept (the older ones only ever meet zero coefficients)."""
    import struct
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
    rng = np.random.default_rng(11)
    n_ch, n1, n2 = 4, 2100, 900
    x = rand_raw(rng, fin, (n_ch, n1 + n2))
    c = rand_raw(rng, fc, (n_taps,)) // 4
    want = OracleFir(n_taps, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch).run(c, x)
    a = A.Fir(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_channels=n_ch, kind="load")
    a.set_coeffs(c)
    a.run(dev(x[:, :n1], fin))
    blob = a.state()
    hl = struct.unpack_from("<Q", blob, 24)[0]                    # StateHdr.per_channel
    assert len(blob) == 64 + n_ch * hl * 2 and hl + 1 >= n_taps
    new_hl = hl + delta
    assert new_hl + 1 >= n_taps
    rows = np.frombuffer(blob[64:], dtype=np.int16).reshape(n_ch, hl)
    hist = x[:, n1 - new_hl:n1].astype(np.int16) if delta > 0 else rows[:, -new_hl:]   # what a handle with that length would hold
    other = blob[:24] + struct.pack("<Q", new_hl) + blob[32:64] + np.ascontiguousarray(hist).tobytes()
    b = A.Fir(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_channels=n_ch, kind="load")
    b.set_coeffs(c)
    b.set_state(other)
    y2 = b.run(dev(x[:, n1:], fin)).cpu().numpy().astype(np.int64)
    assert np.array_equal(y2, want[:, n1:])
    too_short = blob[:24] + struct.pack("<Q", n_taps - 2) + blob[32:64] + b"\0" * (n_ch * (n_taps - 2) * 2)
    with pytest.raises(A.AcdspError):
        b.set_state(too_short)


def test_fir_state_blob_is_checked():
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2)
    a = A.Fir(31, "SHIFT_REG", fin, fc, fa, fo, n_channels=2)
    blob = a.state()
    for other in (A.Fir(33, "SHIFT_REG", fin, fc, fa, fo, n_channels=2), A.Fir(31, "SHIFT_REG", fin, fc, fa, fo, n_channels=3),
                  A.Fir(31, "TRANSPOSED", fin, fc, fa, fo, n_channels=2, kind="load")):
        with pytest.raises(A.AcdspError):
            other.set_state(blob)
    with pytest.raises(A.AcdspError):
        a.set_state(blob[:-2])
    with pytest.raises(A.AcdspError):
        a.set_state(b"XXXXXXXX" + blob[8:])
    a.set_state(blob)


@pytest.mark.parametrize("interp,R,M,N,fin,fo", [(False, 8, 1, 5, A.Fmt(32, 16), A.Fmt(47, 31)), (False, 7, 2, 4, A.Fmt(32, 16), A.Fmt(48, 32)),
                                                  (True, 8, 1, 5, A.Fmt(32, 16), A.Fmt(44, 28)), (True, 7, 2, 5, A.Fmt(16, 8), A.Fmt(33, 25))])
def test_cic_state_round_trip(interp, R, M, N, fin, fo):
    rng = np.random.default_rng(5)
    n_ch, n1, n2 = 4, 1003, 2050          # n1 is not a multiple of R: the decimation phase is part of the state
    x = rand_raw(rng, fin, (n_ch, n1 + n2))
    orc = OracleCic(interp, R, M, N, ofmt(fin), ofmt(fo), n_ch=n_ch)
    w1, w2 = orc.run(x[:, :n1]), orc.run(x[:, n1:])
    a = A.Cic(interp, R, M, N, fin, fo, n_channels=n_ch)
    y1 = a.run(dev(x[:, :n1], fin)).cpu().numpy().astype(np.int64)
    blob = a.state()
    b = A.Cic(interp, R, M, N, fin, fo, n_channels=n_ch)
    b.set_state(blob)
    y2 = b.run(dev(x[:, n1:], fin)).cpu().numpy().astype(np.int64)
    assert np.array_equal(y1, w1) and np.array_equal(y2, w2)
    other = A.Cic(interp, R + 1, M, N, fin, A.Fmt(60, 44), n_channels=n_ch)
    with pytest.raises(A.AcdspError):
        other.set_state(blob)
