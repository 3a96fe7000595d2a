"""Types wider than 64 bits on the GPU (wide.hip through the C ABI) against oracle/acdsp_oracle_wide.cpp.

The reference's templates accept ac_fixed of any width: INT_TYPE of ac_cic_dec_full / ac_cic_intr_full is derived
(include/ac_dsp/ac_cic_dec_full.h:116-137, ac_cic_intr_full.h:107-127) and ACC_TYPE / OUT_TYPE of the FIR classes are the user's
(ac_fir_const_coeffs.h:190-296).  Raw words of such types travel in 16-byte containers (low quadword first)."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from helpers import ofmt
from oracle import OracleCicW, OracleFirW, cic_int_type

pytestmark = pytest.mark.gpu

FT = ["SHIFT_REG", "ROTATE_SHIFT", "C_BUFF", "FOLD_EVEN", "FOLD_ODD", "TRANSPOSED"]
WIDE_FIR = [
    (A.Fmt(32, 16), A.Fmt(32, 16), A.Fmt(96, 48), A.Fmt(96, 48)),
    (A.Fmt(64, 32), A.Fmt(32, 8), A.Fmt(128, 60), A.Fmt(100, 50, True, "RND", "SAT")),
    (A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(72, 44, True, "RND_CONV", "SAT_SYM"), A.Fmt(72, 44)),
    (A.Fmt(40, 20), A.Fmt(24, 4), A.Fmt(80, 30, True, "TRN_ZERO", "WRAP"), A.Fmt(66, 30, True, "RND_INF", "SAT_ZERO")),
    (A.Fmt(20, 10, False), A.Fmt(18, 4), A.Fmt(90, 50, False, "RND", "SAT"), A.Fmt(70, 40, False)),
    (A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(96, 60), A.Fmt(16, 2, True, "RND", "SAT")),     # wide accumulator, narrow output container
]


def rand_raw(rng, f, shape):
    lo = -(1 << (f.W - 1)) if f.S else 0
    hi = (1 << (f.W - 1)) - 1 if f.S else (1 << f.W) - 1
    return rng.integers(lo, hi, size=shape, dtype=np.int64, endpoint=True)


def to_int(y, fmt):
    if A.is_wide(fmt):
        return A.wide_to_int(y)
    v = y.cpu().numpy()
    if not fmt.S:                                  # unsigned raw words come back in signed containers of the same size
        v = v.view({2: np.uint16, 4: np.uint32, 8: np.uint64}[v.dtype.itemsize])
    return v.astype(object)


@pytest.mark.parametrize("ftype", FT)
@pytest.mark.parametrize("k", range(len(WIDE_FIR)))
def test_wide_fir_matches_the_wide_oracle(ftype, k):
    fin, fc, fa, fo = WIDE_FIR[k]
    rng = np.random.default_rng(31 * k + FT.index(ftype))
    for n_taps, n_ch, splits in ((1, 2, [5]), (27, 3, [100, 101, 13]), (64, 2, [300])):
        n = 400
        c = rand_raw(rng, fc, (n_taps,))
        x = rand_raw(rng, fin, (n_ch, n))
        fir = A.Fir(n_taps, ftype, fin, fc, fa, fo, n_channels=n_ch, kind="load")
        fir.set_coeffs(c)
        assert fir.path == "wide"
        xd = torch.from_numpy(x.astype({2: np.int16, 4: np.int32, 8: np.int64}[A.elem_bytes(fin.W)])).cuda()
        cuts = [0] + list(np.cumsum(splits)) + [n]
        got = np.concatenate([to_int(fir.run(xd[:, a:b].contiguous()), fo) for a, b in zip(cuts[:-1], cuts[1:]) if b > a], axis=1)
        want = OracleFirW(n_taps, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch).run(c, x)
        assert np.array_equal(got, want), (ftype, k, n_taps)


def test_wide_transposed_prog_coefficients_change_mid_stream_and_per_channel_sets():
    fin, fc, fa, fo = A.Fmt(32, 16), A.Fmt(24, 8), A.Fmt(100, 50, True, "RND", "SAT"), A.Fmt(80, 40, True, "RND_CONV", "SAT")
    rng = np.random.default_rng(5)
    n_taps, n_ch, n = 19, 3, 240
    c1, c2 = rand_raw(rng, fc, (n_taps,)), rand_raw(rng, fc, (n_taps,))
    x = rand_raw(rng, fin, (n_ch, n))
    xd = torch.from_numpy(x.astype(np.int32)).cuda()
    fir = A.Fir(n_taps, "TRANSPOSED", fin, fc, fa, fo, n_channels=n_ch, kind="prog")
    orc = OracleFirW(n_taps, "TRANSPOSED", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    got, want = [], []
    for cset, (a, b) in ((c1, (0, 100)), (c2, (100, 107)), (c1, (107, n))):
        fir.set_coeffs(cset)
        got.append(to_int(fir.run(xd[:, a:b].contiguous()), fo))
        want.append(orc.run(cset, x[:, a:b]))
    assert np.array_equal(np.concatenate(got, axis=1), np.concatenate(want, axis=1))
    # one coefficient set per channel
    cc = rand_raw(rng, fc, (n_ch, n_taps))
    fir2 = A.Fir(n_taps, "FOLD_ODD", fin, fc, fa, fo, n_channels=n_ch, kind="load", coeffs_per_channel=True)
    fir2.set_coeffs(cc)
    got2 = to_int(fir2.run(xd), fo)
    want2 = OracleFirW(n_taps, "FOLD_ODD", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch).run(cc, x)
    assert np.array_equal(got2, want2)


def test_wide_fir_state_blob_round_trip():
    fin, fc, fa = A.Fmt(32, 16), A.Fmt(24, 8), A.Fmt(100, 50)
    rng = np.random.default_rng(8)
    c = rand_raw(rng, fc, (21,))
    x = rand_raw(rng, fin, (2, 300))
    xd = torch.from_numpy(x.astype(np.int32)).cuda()
    for ftype, kind in (("SHIFT_REG", "load"), ("TRANSPOSED", "prog")):
        a = A.Fir(21, ftype, fin, fc, fa, fa, n_channels=2, kind=kind)
        a.set_coeffs(c)
        y0 = to_int(a.run(xd[:, :130].contiguous()), fa)
        b = A.Fir(21, ftype, fin, fc, fa, fa, n_channels=2, kind=kind)
        b.set_coeffs(c)
        b.set_state(a.state())
        y1 = to_int(b.run(xd[:, 130:].contiguous()), fa)
        want = OracleFirW(21, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fa), n_ch=2).run(c, x)
        assert np.array_equal(np.concatenate([y0, y1], axis=1), want), ftype


WIDE_CIC = [(0, 16, 2, 6, A.Fmt(48, 20)), (1, 16, 2, 6, A.Fmt(48, 20)), (0, 8, 1, 8, A.Fmt(45, 20)), (1, 8, 3, 5, A.Fmt(64, 30)),
            (0, 255, 1, 3, A.Fmt(60, 30, False)), (0, 8, 1, 5, A.Fmt(32, 16))]


@pytest.mark.parametrize("interp,R,M,N,fin", WIDE_CIC)
def test_wide_cic_matches_the_wide_oracle(interp, R, M, N, fin):
    it = cic_int_type(interp, R, M, N, ofmt(fin))
    outs = [A.Fmt(it.W, it.I), A.Fmt(max(it.W - 7, 66), it.I - 3, True, "RND", "SAT"), A.Fmt(100, 50, True, "RND_CONV", "SAT_SYM")]
    if it.W > 64:
        outs.append(A.Fmt(40, 20, True, "RND", "SAT"))      # wide INT_TYPE, narrow output container
    rng = np.random.default_rng(R * N + interp)
    n_ch = 3
    for fo in outs:
        cic = A.Cic(bool(interp), R, M, N, fin, fo, n_channels=n_ch)
        orc = OracleCicW(interp, R, M, N, ofmt(fin), ofmt(fo), n_ch=n_ch)
        for n in ((5 * R + 3, 2, 4 * R) if not interp else (40, 1, 23)):      # several calls: phase and history carry
            x = rand_raw(rng, fin, (n_ch, n))
            xd = torch.from_numpy(x.astype({4: np.int32, 8: np.int64}[A.elem_bytes(fin.W)])).cuda()
            got = to_int(cic.run(xd), fo)
            assert (cic.path == "wide") == (it.W > 64 or fo.W > 64)
            want = orc.run(x)
            assert got.shape == want.shape and np.array_equal(got, want), (interp, R, M, N, n, (fo.W, fo.I))


def test_wide_limits_are_refused_loudly():
    with pytest.raises(A.AcdspError):
        A.Fir(8, "SHIFT_REG", A.Fmt(80, 40), A.Fmt(16, 2), A.Fmt(100, 50), A.Fmt(100, 50))      # IN wider than 64 bits
    with pytest.raises(A.AcdspError):
        A.Fir(8, "SHIFT_REG", A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(130, 50), A.Fmt(100, 50))      # ACC wider than 128 bits
    with pytest.raises(A.AcdspError):
        A.Fir(8, "SHIFT_REG", A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(128, 50, False), A.Fmt(100, 50))   # unsigned 128


def _wide_cases(kind):
    import golden_cases as G
    return G.load((kind,))


def _fmt(a):
    return A.Fmt(a[0], a[1], bool(a[2]), a[3], a[4])


@pytest.mark.parametrize("c", _wide_cases("wide_fir"), ids=lambda c: c["name"])
def test_wide_fir_matches_the_reference_headers_vectors(c):
    """The vectors the reference's own ac_fir_load_coeffs source produced with an 80-bit accumulator (tests/golden/ref_hdr/wide.json)."""
    fin, fc, fa, fo = (_fmt(c[k]) for k in ("in", "coeff", "acc", "out"))
    nch = 3                                              # the same stream on three channels (addressing: the fuzz / sweep tests)
    fir = A.Fir(c["n_taps"], c["ftype"], fin, fc, fa, fo, n_channels=nch, kind="load")
    fir.set_coeffs(np.array(c["coeffs"], dtype=np.int64))
    x = torch.from_numpy(np.tile(np.array(c["x"], dtype=np.int64).astype(np.int32), (nch, 1))).cuda()
    got, pos = [], 0
    for k in c["calls"]:
        got.append(to_int(fir.run(x[:, pos:pos + k].contiguous()), fo))
        pos += k
    got = np.concatenate(got, axis=1)
    want = np.array(c["y"], dtype=object)
    for ch in range(nch):
        assert np.array_equal(got[ch], want), c["name"]


@pytest.mark.parametrize("c", _wide_cases("wide_cic_dec") + _wide_cases("wide_cic_intr"), ids=lambda c: c["name"])
def test_wide_cic_matches_the_reference_headers_vectors(c):
    fin, fo = _fmt(c["in"]), _fmt(c["out"])
    nch = 2
    cic = A.Cic(c["class"] == "wide_cic_intr", c["R"], c["M"], c["N"], fin, fo, n_channels=nch)
    x = torch.from_numpy(np.tile(np.array(c["x"], dtype=np.int64), (nch, 1))).cuda()
    got, pos = [], 0
    for k, want_n in zip(c["calls"], c["outs_per_call"]):
        y = to_int(cic.run(x[:, pos:pos + k].contiguous()), fo)
        assert y.shape[1] == want_n
        got.append(y.reshape(nch, want_n))
        pos += k
    got = np.concatenate(got, axis=1)
    want = np.array(c["y"], dtype=object)
    for ch in range(nch):
        assert np.array_equal(got[ch], want), c["name"]


FUZZ = int(__import__("os").environ.get("ACDSP_FUZZ_CASES", "40"))


@pytest.mark.parametrize("seed", range(FUZZ))
def test_wide_random_types_and_shapes(seed):
    """Seeded differential cases over widths (65 - 128 bit ACC / OUT, any Q / O), tap counts, ftypes, channel counts and call splits."""
    rng = np.random.default_rng(7000 + seed)
    qs, os_ = list(A.Q_MODES), list(A.O_MODES)

    def fmt(wlo, whi, signed=None):
        W = int(rng.integers(wlo, whi + 1))
        S = bool(rng.integers(0, 2)) if signed is None else signed
        if W in (64, 128):
            S = True
        return A.Fmt(W, int(rng.integers(1, W)), S, qs[rng.integers(0, 8)], os_[rng.integers(0, 4)])

    if seed % 3 != 2:
        fin, fc = fmt(4, 40), fmt(4, 32)
        fa, fo = fmt(65, 128), fmt(8, 128)
        ftype = FT[int(rng.integers(0, 6))]
        n_taps, n_ch, n = int(rng.integers(1, 70)), int(rng.integers(1, 5)), int(rng.integers(1, 700))
        c, x = rand_raw(rng, fc, (n_taps,)), rand_raw(rng, fin, (n_ch, n))
        try:
            fir = A.Fir(n_taps, ftype, fin, fc, fa, fo, n_channels=n_ch, kind="prog" if ftype == "TRANSPOSED" else "load")
        except A.AcdspError as e:      # intermediates beyond 256 bits: refused loudly, nothing to compare
            assert "intermediates" in str(e)
            return
        fir.set_coeffs(c)
        assert fir.path == "wide"
        xd = torch.from_numpy(x.astype({2: np.int16, 4: np.int32, 8: np.int64}[A.elem_bytes(fin.W)])).cuda()
        k = int(rng.integers(0, n + 1))
        parts = [to_int(fir.run(xd[:, a:b].contiguous()), fo) for a, b in ((0, k), (k, n)) if b > a]
        want = OracleFirW(n_taps, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch).run(c, x)
        assert np.array_equal(np.concatenate(parts, axis=1), want), (seed, ftype)
    else:
        interp = int(rng.integers(0, 2))
        R, M, N = int(rng.integers(2, 17)), int(rng.integers(1, 4)), int(rng.integers(1, 6))
        if (R * M) ** N >= 2 ** 31:
            N = 2
        fin = A.Fmt(int(rng.integers(34, 65)), int(rng.integers(1, 30)), True)
        it = cic_int_type(interp, R, M, N, ofmt(fin))
        fo = fmt(8, 128) if rng.integers(0, 2) else A.Fmt(it.W, it.I)
        if it.W <= 64 and fo.W <= 64:
            fo = A.Fmt(80, 40, True, "RND", "SAT")
        n_ch = int(rng.integers(1, 4))
        cic = A.Cic(bool(interp), R, M, N, fin, fo, n_channels=n_ch)
        orc = OracleCicW(interp, R, M, N, ofmt(fin), ofmt(fo), n_ch=n_ch)
        for n in (int(rng.integers(1, 300)), int(rng.integers(1, 60))):
            x = rand_raw(rng, fin, (n_ch, n))
            got = to_int(cic.run(torch.from_numpy(x).cuda()), fo)
            want = orc.run(x)
            assert got.shape == want.shape and np.array_equal(got, want), (seed, interp, R, M, N)
