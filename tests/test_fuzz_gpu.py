"""Randomised differential tests on the GPU: engine vs oracle over shapes the targeted tests do not enumerate -- odd burst
lengths, unaligned row starts and strides on either side, call splitting, every kernel family.  Seeds are fixed; raise
ACDSP_FUZZ_CASES for a longer hunt (tools: `ACDSP_FUZZ_CASES=400 python -m pytest tests/test_fuzz_gpu.py -m gpu`)."""
import os

import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from oracle import OracleFir, OracleCic
from helpers import ofmt

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get("ACDSP_FUZZ_CASES", "40"))


def rand_raw(rng, fmt, shape):
    lo = -(1 << (fmt.W - 1)) if fmt.S else 0
    hi = (1 << (fmt.W - 1)) - 1 if fmt.S else (1 << fmt.W) - 1
    return rng.integers(lo, hi + 1, size=shape, dtype=np.int64)


def padded_view(x, dt, lead, tail):
    """device tensor view of x whose rows start `lead` elements into a wider buffer (stride = n + lead + tail)"""
    n_ch, n = x.shape
    big = torch.zeros((n_ch, n + lead + tail), dtype=dt, device="cuda")
    big[:, lead:lead + n] = torch.from_numpy(x).to(dt).cuda()
    return big[:, lead:lead + n]


FIR_TYPES = [
    (A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12)),                       # int8-split MFMA class
    (A.Fmt(16, 1), A.Fmt(16, 1), A.Fmt(44, 14)),
    (A.Fmt(36, 21), A.Fmt(16, 1), A.Fmt(60, 30)),                      # wide words: multi-plane MFMA
    (A.Fmt(24, 8), A.Fmt(18, 2), A.Fmt(50, 12)),
    (A.Fmt(14, 4), A.Fmt(12, 2), A.Fmt(20, 8, True, "RND", "SAT")),    # lossy accumulator: exact-order VALU kernel
    (A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12, False)),                # unsigned wrapping accumulator under signed data
    (A.Fmt(12, 4, False), A.Fmt(14, 1), A.Fmt(36, 12)),                # unsigned input
    # round 5: the families of the R = 1 ring shapes, the class-B matrix-core kernel and the flipped unsigned-16 path
    (A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(64, 32)),                      # the reference const testbench's types
    (A.Fmt(32, 16), A.Fmt(32, 16), A.Fmt(64, 32)),                     # ... load
    (A.Fmt(32, 16), A.Fmt(24, 8), A.Fmt(60, 28)),
    (A.Fmt(28, 6), A.Fmt(23, 7), A.Fmt(64, 32)),                       # ... prog: 6 bits dropped per tap (class B)
    (A.Fmt(28, 6), A.Fmt(23, 7), A.Fmt(60, 31, True, "RND", "WRAP")),  # 7 bits dropped, rounded
    (A.Fmt(16, 8), A.Fmt(24, 6), A.Fmt(48, 26)),                       # class B on 16-bit samples with wide coefficients
    (A.Fmt(16, 2, False), A.Fmt(16, 2), A.Fmt(44, 16)),                # unsigned 16-bit samples: sign-flipped image
    (A.Fmt(16, 2, False), A.Fmt(16, 2), A.Fmt(34, 6)),                 # ... against an accumulator that may wrap
    (A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(24, 8)),                        # class B on 16-bit types (VALU kernel)
]
FIR_OUTS = [A.Fmt(16, 2, True, "RND", "SAT"), A.Fmt(16, 2, True, "TRN", "WRAP"), A.Fmt(16, 6, True, "RND", "SAT"), A.Fmt(40, 12),
            A.Fmt(24, 9, True, "RND", "SAT"), A.Fmt(34, 4, True, "TRN", "WRAP"), A.Fmt(12, 5, True, "RND_CONV", "SAT_SYM"),
            A.Fmt(16, 3, True, "RND_ZERO", "SAT_ZERO"), A.Fmt(16, 2, True, "TRN_ZERO", "SAT_SYM"), A.Fmt(16, 4, True, "RND_INF", "WRAP"), A.Fmt(64, 32)]


@pytest.mark.parametrize("seed", range(CASES))
def test_fir_random_shapes(seed):
    rng = np.random.default_rng(1000 + seed)
    fin, fc, fa = FIR_TYPES[rng.integers(len(FIR_TYPES))]
    fo = FIR_OUTS[rng.integers(len(FIR_OUTS))]
    kind = ["const", "load", "prog", "reg_share"][rng.integers(4)]
    ftypes = ["SHIFT_REG", "FOLD_EVEN", "FOLD_ODD"] + (["FOLD_EVEN_ANTI", "FOLD_ODD_ANTI"] if kind == "reg_share" else ["ROTATE_SHIFT", "C_BUFF", "TRANSPOSED"])
    ftype = ftypes[rng.integers(len(ftypes))]
    n_taps = int(rng.choice([1, 2, 7, 31, 32, 33, 64, 100, 127, 255, 257, 258, 300, 700]))
    if kind == "reg_share" or fa.O:
        n_taps = min(n_taps, 100)                                       # keep the oracle's i128 loops short
    n_ch = int(rng.choice([1, 2, 7, 8, 9, 17]))
    per_ch = bool(rng.integers(2)) and n_taps <= 257 and fin.W <= 16
    n_total = int(rng.choice([1, 5, 100, 1023, 1024, 1025, 2048, 3000, 5000, 9000, 8192, 12288, 10240 + 77]))
    scale = 1 if rng.integers(3) else 8                                 # sometimes small coefficients (zero high-byte planes)
    c = rand_raw(rng, fc, (n_ch, n_taps) if per_ch else (n_taps,)) // scale
    if fc.W > 22:
        c = c >> (fc.W - 22)                                            # three balanced base-256 digits: the matrix-core classes of wide coefficients
    # the fast bodies of every family run WHOLE chunks of aligned rows only (1024 .. 4096 outputs): half of the cases cut the stream at
    # multiples of 1024 and leave the rows aligned, so that the differential test reaches them (round 4's fuzz almost never did)
    whole = bool(rng.integers(2))
    x = rand_raw(rng, fin, (n_ch, n_total))
    fir = A.Fir(n_taps, ftype, fin, fc, fa, fo, n_channels=n_ch, kind=kind, coeffs_per_channel=per_ch)
    fir.set_coeffs(c)
    orc = OracleFir(n_taps, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch, reg_share=(1, 1, 0) if kind == "reg_share" else None)
    yo = orc.run(c, x)
    cuts = sorted(set(int(v) for v in rng.integers(0, n_total + 1, size=rng.integers(0, 3))))
    if whole:
        cuts = sorted(set(min(n_total, (v // 1024) * 1024) for v in cuts))
    bounds = [0] + cuts + [n_total]
    dt_in, dt_out = A.torch_dtype_for(fin), A.torch_dtype_for(fo)
    outs = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        if b == a:
            continue
        pad = 8 if whole else 1                                         # whole: rows start on 16-byte boundaries, strides are multiples of 16 bytes
        xv = padded_view(x[:, a:b], dt_in, pad * int(rng.integers(0, 9 // pad + 1)), pad * int(rng.integers(0, 9 // pad + 1)) + ((-(b - a)) % 8 if whole else 0))
        out = torch.zeros((n_ch, b - a + (((-(b - a)) % 8) if whole else int(rng.integers(0, 9)))), dtype=dt_out, device="cuda")
        if not whole:
            out = out[:, int(rng.integers(0, min(4, out.shape[1] - (b - a) + 1))):]
        y = fir.run(xv, out)[:, :b - a].cpu().numpy().astype(np.int64)
        outs.append(y)
    y = np.concatenate(outs, axis=1)
    bad = np.argwhere(y != yo)
    assert bad.size == 0, "seed %d: %d mismatches, first %s (kernel %s, %s %s taps %d ch %d n %d cuts %s per_ch %s whole %s types %s)" % (
        seed, len(bad), bad[0], fir.kernel, kind, ftype, n_taps, n_ch, n_total, cuts, per_ch, whole, (fin.W, fc.W, fa.W, fo.W))


@pytest.mark.parametrize("seed", range(max(CASES // 2, 1)))   # the oracle needs ~1 s per case at these lengths
def test_fir_long_band_limited_sets(seed):
    """258 .. 1025 taps, low-pass sets of random cutoff and gain (the register-resident shapes for 11 .. 33 K-blocks when the high-byte
    band is narrow, the LDS-resident kernels when it is not), one set or a set per channel, ragged calls on unaligned views."""
    from helpers import windowed_sinc
    rng = np.random.default_rng(7000 + seed)
    fin, fc, fa = FIR_TYPES[rng.integers(2)]
    fo = FIR_OUTS[rng.integers(4)]
    kind = ["const", "load", "prog"][rng.integers(3)]
    ftype = ["SHIFT_REG", "C_BUFF", "FOLD_ODD", "ROTATE_SHIFT"][rng.integers(4)]
    n_taps = int(rng.integers(258, 1026))
    if ftype == "FOLD_ODD":
        n_taps |= 1
    n_ch = int(rng.choice([1, 3, 8, 9]))
    per_ch = bool(rng.integers(2))
    n_total = int(rng.choice([700, 2048, 5000, 8192 + 40, 20000]))
    def one():
        return windowed_sinc(n_taps, float(rng.uniform(0.01, 0.2)), fc, gain=float(rng.choice([0.3, 0.9, 1.0, 3.0])))
    c = np.stack([one() for _ in range(n_ch)]) if per_ch else one()
    x = rand_raw(rng, fin, (n_ch, n_total))
    fir = A.Fir(n_taps, ftype, fin, fc, fa, fo, n_channels=n_ch, kind=kind, coeffs_per_channel=per_ch)
    fir.set_coeffs(c)
    yo = OracleFir(n_taps, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch).run(c, x)
    cuts = sorted(set(int(v) for v in rng.integers(0, n_total + 1, size=rng.integers(0, 3))))
    bounds = [0] + cuts + [n_total]
    dt_in, dt_out = A.torch_dtype_for(fin), A.torch_dtype_for(fo)
    outs = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        if b == a:
            continue
        xv = padded_view(x[:, a:b], dt_in, int(rng.integers(0, 9)), int(rng.integers(0, 9)))
        out = torch.zeros((n_ch, b - a + int(rng.integers(0, 9))), dtype=dt_out, device="cuda")
        outs.append(fir.run(xv, out)[:, :b - a].cpu().numpy().astype(np.int64))
    y = np.concatenate(outs, axis=1)
    bad = np.argwhere(y != yo)
    assert bad.size == 0, "seed %d: %d mismatches, first %s (path %s issued %d, %s %s taps %d ch %d n %d cuts %s per_ch %s)" % (
        seed, len(bad), bad[0], fir.path, fir.mfma_issued(), kind, ftype, n_taps, n_ch, n_total, cuts, per_ch)


@pytest.mark.parametrize("seed", range(CASES))
def test_cic_random_shapes(seed):
    rng = np.random.default_rng(2000 + seed)
    interp = bool(rng.integers(2))
    R = int(rng.choice([2, 3, 5, 6, 7, 8, 10, 12, 16, 20, 32, 36, 40, 42, 48, 49, 64, 100]))   # from 32: the two-stage decimator where a compiled rate divides
    M = int(rng.choice([1, 2, 3]))
    N = int(rng.choice([1, 2, 4, 5]))
    fin = [A.Fmt(16, 1), A.Fmt(32, 16), A.Fmt(12, 12, False), A.Fmt(20, 4), A.Fmt(24, 8)][rng.integers(5)]
    try:
        probe = A.Cic(interp, R, M, N, fin, fin, n_channels=1)
    except A.AcdspError:
        pytest.skip("outside the reference's own limits")
    it = probe.int_type
    fout = [A.Fmt(it.W, it.I), A.Fmt(it.W, it.I), A.Fmt(20, 14, True, "RND", "SAT"), A.Fmt(16, it.I - (it.W - 16), True, "TRN", "WRAP")][rng.integers(4)]
    if fout.W > 64 or it.W > 64:
        pytest.skip("wider than 64 bits")
    n_ch = int(rng.choice([1, 3, 64, 65]))
    n_total = int(rng.choice([1, 17, 256, 1000, 4096, 20000, 16 * R * 256])) if not interp else int(rng.choice([1, 2, 17, 300, 2500]))
    x = rand_raw(rng, fin, (n_ch, n_total))
    cic = A.Cic(interp, R, M, N, fin, fout, n_channels=n_ch)
    orc = OracleCic(interp, R, M, N, ofmt(fin), ofmt(fout), n_ch=n_ch)
    cuts = sorted(set(int(v) for v in rng.integers(0, n_total + 1, size=rng.integers(0, 3))))
    bounds = [0] + cuts + [n_total]
    dt = A.torch_dtype_for(fin)
    ys, yos = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        if b == a:
            continue
        lead = int(rng.integers(0, 5)) * (0 if rng.integers(2) else 1)
        xv = padded_view(x[:, a:b], dt, lead, int(rng.integers(0, 20)))
        ys.append(cic.run(xv).cpu().numpy().astype(np.int64))
        yos.append(orc.run(x[:, a:b]))
    y, yo = np.concatenate(ys, axis=1), np.concatenate(yos, axis=1)
    assert y.shape == yo.shape, (y.shape, yo.shape)
    bad = np.argwhere(y != yo)
    assert bad.size == 0, "seed %d: %d mismatches, first %s (interp %s R %d M %d N %d ch %d n %d cuts %s)" % (
        seed, len(bad), bad[0], interp, R, M, N, n_ch, n_total, cuts)


@pytest.mark.parametrize("seed", range(max(CASES // 4, 8)))
def test_ddc_random_bursts(seed):
    # fused cascade (config-5 shape class): bursts of any length (whole 16-sample slots readable: rows are padded), any split
    from test_ddc_gpu import oracle_cascade
    from test_fir_gpu import windowed_sinc
    rng = np.random.default_rng(3000 + seed)
    cin, fc, fa = A.Fmt(16, 1), A.Fmt(16, 1), A.Fmt(60, 30)
    fo = [A.Fmt(24, 9, True, "RND", "SAT"), A.Fmt(32, 12, True, "TRN", "WRAP"), A.Fmt(20, 6, True, "RND", "SAT")][rng.integers(3)]
    n_taps = int(rng.choice([127, 101, 64]))
    n_ch = int(rng.choice([1, 2, 5]))
    n_total = int(rng.choice([40, 1000, 16 * 300 + 5, 16 * 1024 * 3 + 77, 50000]))
    x = rand_raw(rng, cin, (n_ch, n_total))
    c = windowed_sinc(n_taps, 0.2, fc)
    ddc = A.Ddc(16, 1, 5, cin, n_taps, "SHIFT_REG", fc, fa, fo, n_channels=n_ch)
    ddc.set_coeffs(c)
    cuts = sorted(set(int(v) for v in rng.integers(0, n_total + 1, size=rng.integers(0, 3))))
    bounds = [0] + cuts + [n_total]
    ys = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        if b == a:
            continue
        w = (b - a + 15) // 16 * 16 + 16 * int(rng.integers(0, 2))
        big = torch.zeros((n_ch, w), dtype=torch.int16, device="cuda")
        big[:, :b - a] = torch.from_numpy(x[:, a:b]).to(torch.int16).cuda()
        ys.append(ddc.run(big[:, :b - a]).cpu().numpy().astype(np.int64))
    y = np.concatenate(ys, axis=1)
    yo = oracle_cascade(16, 1, 5, cin, ddc.int_type, n_taps, "SHIFT_REG", fc, fa, fo, c, x, cuts)
    assert y.shape == yo.shape
    bad = np.argwhere(y != yo)
    assert bad.size == 0, "seed %d: %d mismatches, first %s (path %s taps %d n %d cuts %s)" % (seed, len(bad), bad[0], ddc.path, n_taps, n_total, cuts)


@pytest.mark.parametrize("seed", range(max(CASES // 2, 10)))
def test_intg_dump_random_block_sequences(seed):
    # mixes dumping blocks with n_sample = 0 / > NS blocks across calls: tiled kernel, general kernel and the carried sums
    from oracle import OracleIntgDump
    rng = np.random.default_rng(4000 + seed)
    ns = int(rng.choice([4, 64, 100, 1024]))
    chn = int(rng.choice([1, 2, 3, 4, 8, 5, 6, 7, 12, 16]))   # 3, 5, 6, 7, 12, 16: the matrix-core de-interleave when every block dumps after NS rounds
    fin = [A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(12, 12, False)][rng.integers(3)]
    fa, fo = [(A.Fmt(40, 20), A.Fmt(40, 20)), (A.Fmt(20, 10, True, "TRN", "SAT"), A.Fmt(12, 8, True, "RND", "SAT")),
              (A.Fmt(48, 40, False), A.Fmt(48, 40, False)), (A.Fmt(64, 32), A.Fmt(64, 32)), (A.Fmt(64, 32), A.Fmt(32, 12, True, "TRN", "SAT"))][rng.integers(5)]
    n_obj = int(rng.choice([1, 5]))
    eng = A.IntgDump(ns, chn, fin, fa, fo, n_objects=n_obj)
    orc = OracleIntgDump(ns, chn, ofmt(fin), ofmt(fa), ofmt(fo), n_obj=n_obj)
    for _ in range(int(rng.integers(1, 5))):
        nb = int(rng.integers(1, 200)) if ns < 1024 else int(rng.integers(1, 24))
        n_sample = rng.integers(1, ns + 1, size=nb)
        if rng.integers(3) == 0:                      # every block dumps after NS rounds: the streaming / matrix-core kernels' shape
            n_sample[:] = ns
        elif rng.integers(2):
            n_sample[rng.integers(0, nb, size=max(1, nb // 10))] = rng.choice([0, ns + 5])
        ni, no = eng.counts(n_sample)
        x = rand_raw(rng, fin, (n_obj, ni))
        y = eng.run(torch.from_numpy(x).to(A.torch_dtype_for(fin)).cuda(), n_sample).cpu().numpy().astype(np.int64)
        assert np.array_equal(y, orc.run(x, n_sample)), seed


@pytest.mark.parametrize("seed", range(max(CASES // 2, 10)))
def test_poly_dec_and_intr_random_shapes(seed):
    from oracle import OraclePolyDec, OraclePolyIntr
    rng = np.random.default_rng(5000 + seed)
    lossless = bool(rng.integers(2))
    fin, fc = (A.Fmt(16, 2), A.Fmt(16, 2)) if rng.integers(2) else (A.Fmt(20, 6), A.Fmt(14, 3))
    fa = A.Fmt(fin.W + fc.W + 9, fin.I + fc.I + 9) if lossless else A.Fmt(22, 9, True, ["TRN", "RND", "RND_CONV"][rng.integers(3)], ["WRAP", "SAT"][rng.integers(2)])
    fo = [A.Fmt(16, 2, True, "RND", "SAT"), fa, A.Fmt(18, 7, True, "TRN", "WRAP")][rng.integers(3)]
    n_ch = int(rng.choice([1, 3, 9]))
    # ---- decimator
    nt, df = int(rng.choice([1, 3, 8, 16])), int(rng.choice([1, 2, 5, 8, 16]))
    c = rand_raw(rng, fc, (nt * df,))
    dec = A.PolyDec(nt, df, fin, fc, fa, fo, n_channels=n_ch)
    dec.set_coeffs(c)
    od = OraclePolyDec(nt, df, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    for groups in (int(rng.integers(1, 40)), int(rng.choice([16, 704, 2048 + 16 * int(rng.integers(0, 9))]))):
        x = rand_raw(rng, fin, (n_ch, groups * df))
        y = dec.run(torch.from_numpy(x).to(A.torch_dtype_for(fin)).cuda()).cpu().numpy().astype(np.int64)
        assert np.array_equal(y, od.run(c, x)), ("dec", seed, dec.path)
    # ---- interpolator
    ftype = ["FOLD_EVEN", "FOLD_ODD", "FOLD_ANTI"][rng.integers(3)]
    n_taps, ifac = int(rng.choice([2, 5, 8, 15, 16])), int(rng.choice([1, 2, 3, 8]))
    j = ifac - 1
    csz = {"FOLD_EVEN": (n_taps // 2 - 1) + j * n_taps // 2, "FOLD_ODD": (n_taps - 1) // 2 + (n_taps // 2 + 1) * j,
           "FOLD_ANTI": (n_taps - 1) + n_taps * j}[ftype] + 1 + int(rng.integers(0, 3))
    if csz < 1:
        csz = 1
    ci = rand_raw(rng, fc, (csz,))
    sign = rng.integers(0, 2, size=ifac)
    corr = rng.permutation(ifac) if rng.integers(2) else np.arange(ifac)
    pi = A.PolyIntr(n_taps, csz, ifac, ftype, fin, fc, fa, fo, n_channels=n_ch)
    pi.set_ctrl(ci, sign, corr)
    oi = OraclePolyIntr(n_taps, csz, ifac, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    for n in (1, int(rng.integers(1, 50)), int(rng.choice([300, 1500]))):
        x = rand_raw(rng, fin, (n_ch, n))
        y = pi.run(torch.from_numpy(x).to(A.torch_dtype_for(fin)).cuda()).cpu().numpy().astype(np.int64)
        yo = oi.run(ci, sign, corr, x)
        assert y.shape == yo.shape and np.array_equal(y, yo), ("intr", seed, ftype, n_taps, ifac)


MV_ACC = [A.Fmt(40, 18), A.Fmt(48, 20), A.Fmt(60, 30), A.Fmt(56, 36, True, "RND"), A.Fmt(32, 16), A.Fmt(32, 16, True, "RND"), A.Fmt(30, 14, True, "RND"), A.Fmt(36, 14, False),
          A.Fmt(24, 4), A.Fmt(40, 18, True, "RND_CONV", "SAT")]
MV_OUT = [A.Fmt(16, 8, True, "RND", "SAT"), A.Fmt(64, 32), A.Fmt(32, 16, True, "RND", "SAT"), A.Fmt(24, 10, True, "TRN", "WRAP"), A.Fmt(40, 18), A.Fmt(12, 6, False, "RND", "SAT"),
          A.Fmt(33, 20, True, "RND", "SAT"), A.Fmt(16, 8, True, "RND_INF", "SAT_SYM"), A.Fmt(9, 3, False, "TRN", "WRAP")]


@pytest.mark.parametrize("seed", range(CASES))
def test_mv_avg_random_shapes(seed):
    """ac_mv_avg: streaming kernel (aligned frames, small weights) and the general kernels, picked by the shapes / types drawn."""
    from oracle import OracleMvAvg
    rng = np.random.default_rng(6000 + seed)
    fin = [A.Fmt(16, 8), A.Fmt(14, 3), A.Fmt(15, 8, False), A.Fmt(16, 8, False), A.Fmt(24, 12), A.Fmt(32, 16), A.Fmt(28, 9, False)][rng.integers(7)]
    fc = [A.Fmt(16, 2), A.Fmt(12, 1), A.Fmt(16, 16), A.Fmt(10, 0, False), A.Fmt(24, 2)][rng.integers(5)]
    fa, fo = MV_ACC[rng.integers(len(MV_ACC))], MV_OUT[rng.integers(len(MV_OUT))]
    taps = int(rng.choice([1, 3, 5, 7, 9, 11, 15, 17, 23, 25, 31, 33, 35, 41, 49, 65]))
    mode = ["WIN", "MIRROR", "CLIP"][rng.integers(3)]
    n_sample = int(rng.choice([8, 16, 24, 64, 128, 200, 256, 504, 512, 520, 1000, 1024, 1032, 2056]))
    n_frames, n_obj = int(rng.choice([1, 2, 3, 8, 16, 17])), int(rng.choice([1, 2, 5]))
    x = rand_raw(rng, fin, (n_obj, n_sample * n_frames))
    c = rand_raw(rng, fc, (taps,))
    if rng.integers(3):                                   # mostly: weights small enough for the int32 class
        c = c // int(max(1, (np.abs(c).sum() >> 14) + 1))
    lead = int(rng.choice([0, 0, 8, 3]))
    eng = A.MvAvg(4096, taps, mode, fin, fc, fa, fo, n_objects=n_obj)
    eng.set_coeffs(c)
    xv = padded_view(x, A.torch_dtype_for(fin), lead, int(rng.integers(0, 9)))
    y = eng.run(xv, n_sample).cpu().numpy().astype(np.int64)
    yo = OracleMvAvg(taps, mode, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_obj=n_obj).run(c, x, n_sample)
    assert y.shape == yo.shape, (y.shape, yo.shape)
    bad = np.argwhere(y != yo)
    assert bad.size == 0, "seed %d: %d mismatches, first %s (path %s, taps %d %s n %d x %d obj %d, %s %s %s %s)" % (
        seed, len(bad), bad[0], eng.path, taps, mode, n_sample, n_frames, n_obj, fin, fc, fa, fo)
