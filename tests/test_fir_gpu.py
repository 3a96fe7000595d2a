"""GPU parity tests, FIR: the HIP engine (through the C ABI) against the CPU oracle, bit for bit."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from oracle import OracleFir, stimulus
from helpers import ofmt, two_tone, read_fracs, sqnr_db, windowed_sinc
from oracle import from_double

pytestmark = pytest.mark.gpu

FTYPES6 = ["SHIFT_REG", "ROTATE_SHIFT", "C_BUFF", "FOLD_EVEN", "FOLD_ODD", "TRANSPOSED"]


def rand_raw(rng, fmt, shape):
    lo = -(1 << (fmt.W - 1)) if fmt.S else 0
    hi = (1 << (fmt.W - 1)) - 1 if fmt.S else (1 << fmt.W) - 1
    return rng.integers(lo, hi + 1, size=shape, dtype=np.int64)


def run_engine(fir, x, splits=None):
    """x: [n_ch][n] int64 raw -> int64 raw, through device tensors; optional call splitting."""
    dt = A.torch_dtype_for(fir.fin)
    outs = []
    bounds = [0] + list(splits or []) + [x.shape[1]]
    for a, b in zip(bounds[:-1], bounds[1:]):
        if b == a:
            continue
        xd = torch.from_numpy(x[:, a:b].copy()).to(dt).cuda()
        outs.append(fir.run(xd).cpu().numpy().astype(np.int64))
    return np.concatenate(outs, axis=1)


def check_case(n_taps, ftype, fin, fc, fa, fo, n_ch=3, n=700, kind="load", per_channel=False, splits=None, seed=0,
               force_generic=False, coeffs=None, expect_path=None):
    rng = np.random.default_rng(seed)
    x = rand_raw(rng, fin, (n_ch, n))
    if coeffs is None:
        coeffs = rand_raw(rng, fc, (n_ch, n_taps) if per_channel else (n_taps,))
    fir = A.Fir(n_taps, ftype, fin, fc, fa, fo, n_channels=n_ch, kind=kind, coeffs_per_channel=per_channel,
                force_generic=force_generic)
    fir.set_coeffs(coeffs)
    if expect_path:
        assert fir.path == expect_path, fir.path
    y = run_engine(fir, x, splits)
    orc = OracleFir(n_taps, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    yo = orc.run(coeffs, x)
    bad = np.argwhere(y != yo)
    assert bad.size == 0, "%d mismatches, first at %s: got %d want %d (path %s)" % (
        len(bad), bad[0], y[tuple(bad[0])], yo[tuple(bad[0])], fir.path)
    return fir


@pytest.mark.parametrize("ftype", FTYPES6)
def test_reference_test_types_all_ftypes(ftype):
    # types of tests/rtest_ac_fir_load_coeffs.cpp:50-74 (lossless 64-bit accumulator)
    check_case(27, ftype, A.Fmt(32, 16), A.Fmt(32, 16), A.Fmt(64, 32), A.Fmt(64, 32), splits=[100, 113, 400])


@pytest.mark.parametrize("ftype", FTYPES6)
def test_prog_test_types_lossy_acc(ftype):
    # tests/rtest_ac_fir_prog_coeffs.cpp:47-54: F_in+F_c = 38 > F_acc = 32 -> per-tap truncation
    check_case(27, ftype, A.Fmt(28, 6), A.Fmt(23, 7), A.Fmt(64, 32), A.Fmt(64, 32), kind="prog", splits=[1, 2, 30])


@pytest.mark.parametrize("q", list(A.Q_MODES))
@pytest.mark.parametrize("o", list(A.O_MODES))
def test_all_q_o_modes_in_accumulator_and_output(q, o):
    # sign/parity dependent rounding and saturation make the tap order observable
    for ftype in ("SHIFT_REG", "C_BUFF", "FOLD_ODD", "TRANSPOSED"):
        check_case(9, ftype, A.Fmt(12, 4), A.Fmt(10, 2), A.Fmt(18, 7, True, q, o), A.Fmt(9, 5, True, q, o), n=300,
                   seed=hash((q, o)) % 1000, splits=[5])


@pytest.mark.parametrize("q", list(A.Q_MODES))
@pytest.mark.parametrize("o", list(A.O_MODES))
def test_all_q_o_modes_of_a_16_bit_output_on_the_matrix_core_epilogue(q, o):
    """Every rounding / overflow mode of a full 16-bit OUT_TYPE behind an exact accumulator: the 32-bit epilogue of the int8 kernel with the
    increment of the dropped bits (round 5; these ran the generic epilogue with element-wise stores at 0.12 of the roofline).  Whole
    1024-sample steps (the software-pipelined body) and ragged calls (the edge body), shifts on both sides of 16, gains that saturate, values
    on the rounding ties (samples and coefficients that are multiples of large powers of two), 63 and 255 taps."""
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(42, 14)
    for k, (n_taps, fo) in enumerate(((63, A.Fmt(16, 2, True, q, o)), (255, A.Fmt(16, 3, True, q, o)), (31, A.Fmt(16, 9, True, q, o)), (127, A.Fmt(16, 0, True, q, o)))):
        rng = np.random.default_rng(n_taps)
        c = windowed_sinc(n_taps, 0.1, fc) if k != 3 else (rand_raw(rng, fc, (n_taps,)) >> 4)
        if k == 2:
            c = (c >> 6) << 6                               # products that land on the ties of the dropped bits
        x = rand_raw(rng, fin, (3, 5 * 1024 + 77))
        x[1, :] = (x[1, :] >> 9) << 9
        x[2, 1000:3000] = 32767
        fir = A.Fir(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_channels=3)
        fir.set_coeffs(c)
        assert fir.path == "mfma_i8", fir.path
        y = run_engine(fir, x, [1024 + 16, 4096 + 16])     # 1040 (edge body), 3072 = three whole steps (pipelined body), the ragged rest
        yo = OracleFir(n_taps, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=3).run(c, x)
        bad = np.argwhere(y != yo)
        assert bad.size == 0, "%s %s case %d: %d mismatches, first at %s: got %d want %d" % (q, o, k, len(bad), bad[0], y[tuple(bad[0])], yo[tuple(bad[0])])


@pytest.mark.parametrize("types", [(A.Fmt(8, 1), A.Fmt(8, 1), A.Fmt(30, 15), A.Fmt(8, 1, True, "RND", "SAT")),      # (accumulators that hold 255 full-range taps)
                                   (A.Fmt(8, 1), A.Fmt(8, 1), A.Fmt(30, 15), A.Fmt(8, 1, True, "TRN", "WRAP")),
                                   (A.Fmt(10, 2), A.Fmt(8, 2), A.Fmt(32, 16), A.Fmt(10, 2, True, "RND", "SAT")),
                                   (A.Fmt(8, 4), A.Fmt(6, 1), A.Fmt(30, 18), A.Fmt(12, 4, True, "RND", "SAT")),
                                   (A.Fmt(8, 1), A.Fmt(8, 1), A.Fmt(30, 16), A.Fmt(16, 2, True, "RND_CONV", "SAT"))])   # rs = 0 into 16 bits: every product bit kept
@pytest.mark.parametrize("n_taps", [15, 63, 255])
def test_narrow_types_run_the_fast_epilogue_on_pre_scaled_coefficients(types, n_taps):
    """ADC-width filters end to end (<8,1> samples, coefficients and output: 7 dropped bits): less than one bit is left to shift by in the 32-bit
    epilogue once the packed finish of a narrow OUT_TYPE has taken its 16 - W bits, so the coefficient set goes through the kernel scaled by a
    power of two (engine_fir.hip: mfma_cshift).  Whole steps (pipelined body) and ragged calls, full-range sets, a set per channel."""
    fin, fc, fa, fo = types
    rng = np.random.default_rng(n_taps + fo.W)
    c = rand_raw(rng, fc, (n_taps,))
    fir = check_case(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_ch=3, n=5 * 1024 + 99, splits=[1024 + 8, 4096 + 8], seed=n_taps, coeffs=c, expect_path="mfma_i8")
    epi, cshift, _ = fir.mfma_epilogue()
    assert epi in (1, 2) and cshift >= 1, (epi, cshift)
    check_case(n_taps, "FOLD_ODD" if n_taps % 2 else "FOLD_EVEN", fin, fc, fa, fo, n_ch=4, n=3 * 1024, per_channel=n_taps <= 63, seed=n_taps + 1,
               coeffs=(rand_raw(rng, fc, (4, n_taps)) >> 1) if n_taps <= 63 else (c >> 1), expect_path="mfma_i8")


def test_unsigned_types():
    check_case(8, "FOLD_EVEN", A.Fmt(10, 3, False), A.Fmt(9, 1, False), A.Fmt(24, 8, False), A.Fmt(12, 6, False, "RND", "SAT"))
    check_case(8, "SHIFT_REG", A.Fmt(10, 3, False), A.Fmt(9, 1, True), A.Fmt(24, 8, True), A.Fmt(12, 6, False, "RND", "SAT"))


@pytest.mark.parametrize("ftype", ["FOLD_EVEN", "FOLD_ODD"])
@pytest.mark.parametrize("n_taps", [4, 5, 6, 7])
def test_fold_parity_quirks(ftype, n_taps):
    # FOLD_EVEN on an odd tap count drops the centre tap, FOLD_ODD on an even one ignores reg[N/2]
    check_case(n_taps, ftype, A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT"))


def test_per_channel_coefficients():
    check_case(33, "SHIFT_REG", A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(40, 12), n_ch=5, per_channel=True)
    check_case(12, "FOLD_EVEN", A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(20, 4), A.Fmt(16, 2), n_ch=5, per_channel=True)


def test_per_channel_coefficients_on_the_mfma_path():
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12)
    rng = np.random.default_rng(77)
    c = np.minimum(rand_raw(rng, fc, (9, 200)), 32639)
    check_case(200, "SHIFT_REG", fin, fc, fa, A.Fmt(16, 2, True, "RND", "SAT"), n_ch=9, n=3000, per_channel=True, coeffs=c,
               expect_path="mfma_i8", splits=[1000, 1024, 2047])
    check_case(200, "C_BUFF", fin, fc, fa, fa, n_ch=9, n=1500, per_channel=True, coeffs=c, expect_path="mfma_i8")


def test_transposed_coefficient_reload_mid_stream():
    # reg_trans[] keeps partial sums made with the coefficients of their time (ac_fir_load_coeffs.h:265-278)
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(24, 5, True, "RND_CONV", "SAT"), A.Fmt(16, 3)
    rng = np.random.default_rng(5)
    n_ch, N = 2, 13
    fir = A.Fir(N, "TRANSPOSED", fin, fc, fa, fo, n_channels=n_ch, kind="load")
    orc = OracleFir(N, "TRANSPOSED", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    for n in (40, 3, 1, 25):
        c = rand_raw(rng, fc, (N,))
        x = rand_raw(rng, fin, (n_ch, n))
        fir.set_coeffs(c)
        y = run_engine(fir, x)
        assert np.array_equal(y, orc.run(c, x))


@pytest.mark.parametrize("N,per_channel", [(13, False), (63, False), (255, False), (64, True), (300, False)])
def test_transposed_reload_in_the_exact_sum_class_runs_the_matrix_cores(N, per_channel):
    """TRANSPOSED with loadable coefficients and an accumulator that cannot lose bits: the n_taps - 1 outputs behind a coefficient change
    come from reg_trans on the exact-order kernel, everything else from the input history on the matrix cores (engine.hip: rt_hybrid).
    Reloads closer together than n_taps - 1 samples, calls of one sample, the same set handed over again, state blobs taken inside a
    transition and in the steady state, all against the oracle's reg_trans recurrence."""
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(42, 14), A.Fmt(16, 2, True, "RND", "SAT")
    rng = np.random.default_rng(N)
    n_ch = 5
    mk = lambda: A.Fir(N, "TRANSPOSED", fin, fc, fa, fo, n_channels=n_ch, kind="load", coeffs_per_channel=per_channel)
    fir, other = mk(), None
    orc = OracleFir(N, "TRANSPOSED", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    c = None
    lens = (3 * N + 40, 5, max(N // 2, 1), 1, N - 1 if N > 1 else 1, 2 * N + 17, N // 3 + 1, 4 * N, 16, 2 * N)
    new_set = (True, True, True, False, True, False, True, True, False, False)
    for i, (n, change) in enumerate(zip(lens, new_set)):
        if change or c is None:
            c = np.minimum(rand_raw(rng, fc, (n_ch, N) if per_channel else (N,)), 32639)
        fir.set_coeffs(c)                                  # unchanged sets are handed over again, as ac_fir_prog_coeffs does
        if other is not None:
            other.set_coeffs(c)
        x = rand_raw(rng, fin, (n_ch, n))
        want = orc.run(c, x)
        assert np.array_equal(run_engine(fir, x), want), "call %d" % i
        assert fir.path == "mfma_i8"
        if other is not None:
            assert np.array_equal(run_engine(other, x), want), "call %d after a state load" % i
            other = None
        if i in (1, 2, 5, 8):                              # 1, 2: inside a transition; 5, 8: steady state (reg_trans rebuilt from the history)
            other = mk()
            other.set_coeffs(c)
            other.set_state(fir.state())


@pytest.mark.parametrize("n_taps", [15, 63, 127, 255, 300, 600])
@pytest.mark.parametrize("fo", [A.Fmt(12, 1, True, "RND", "SAT"), A.Fmt(14, 2, True, "TRN", "SAT"), A.Fmt(8, 1, True, "RND", "SAT"),
                                A.Fmt(12, 1, True, "RND", "WRAP"), A.Fmt(15, 2, True, "TRN", "WRAP"), A.Fmt(9, 3, True, "RND", "WRAP"),
                                A.Fmt(2, 1, True, "RND", "SAT")])
def test_output_types_of_fewer_than_16_bits_keep_the_32_bit_epilogue(n_taps, fo):
    """ADC-width OUT_TYPEs (<12,..>, <14,..>, ...): AC_SAT clamps to the narrower range, AC_WRAP sign-extends the low W bits, both on the packed
    tile words of the fast epilogue classes (up to 9 K-blocks; longer filters take the 64-bit branch-free epilogue).  Windowed-sinc sets
    (band-limited high byte plane: the register-resident kernels) and dense random ones, unit gain and a gain that saturates."""
    from bench import windowed_sinc_raw
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(42, 14)
    for k, c in enumerate((windowed_sinc_raw(n_taps | 1, 0.1, fc.F)[:n_taps],
                           np.minimum(rand_raw(np.random.default_rng(n_taps), fc, (n_taps,)), 32639) >> (6 if n_taps > 255 else 3))):
        # calls of at least two complete 1024-sample steps: the software-pipelined body (round 4's sizes -- 1040 + 1008 + ... samples -- only
        # ever ran the edge body, and the pipelined write-out of these types stored every lane's first dword four times: round 5)
        fir = check_case(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_ch=5, n=6 * 1024 + 3 * n_taps + 8, splits=[1024 + 16, 4096 + 16], seed=n_taps + fo.W + k,
                         coeffs=np.asarray(c, dtype=np.int64), expect_path="mfma_i8")
        del fir


@pytest.mark.parametrize("n_taps", [15, 63, 127, 255, 300])
@pytest.mark.parametrize("fo", [A.Fmt(24, 6, True, "RND", "SAT"), A.Fmt(32, 12, True, "TRN", "WRAP"), A.Fmt(20, 4, True, "RND", "WRAP"),
                                A.Fmt(32, 3, True, "RND", "SAT"), A.Fmt(18, 2, True, "TRN", "SAT"), A.Fmt(30, 2, True, "TRN", "WRAP")])
def test_output_types_of_17_to_32_bits_take_the_wide_class_with_an_int32_tile(n_taps, fo):
    """4-byte OUT containers: the wide (64-bit recombination) class of the pipelined kernel with a 4 KB int32 tile, AC_WRAP and AC_SAT; the
    chunk edges and filters past 9 K-blocks on the generic epilogue.  Fraction counts from rs = 2 (fo.I = 2: 30 fraction bits kept) to 24."""
    from bench import windowed_sinc_raw
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(42, 14)
    for k, c in enumerate((windowed_sinc_raw(n_taps | 1, 0.1, fc.F)[:n_taps],
                           np.minimum(rand_raw(np.random.default_rng(n_taps), fc, (n_taps,)), 32639) >> 2)):
        check_case(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_ch=5, n=8 * 1024 + 3 * n_taps + 8, splits=[2048 + 16], seed=n_taps + fo.W + k,
                   coeffs=np.asarray(c, dtype=np.int64), expect_path="mfma_i8")


@pytest.mark.parametrize("n_taps", [1, 7, 8, 63, 255, 300])
@pytest.mark.parametrize("ftype", ["SHIFT_REG", "ROTATE_SHIFT", "C_BUFF"])
@pytest.mark.parametrize("fa", [A.Fmt(24, 8), A.Fmt(24, 8, True, "RND", "WRAP"), A.Fmt(40, 13, True, "RND", "WRAP"), A.Fmt(20, 3, False, "TRN", "WRAP"),
                                A.Fmt(44, 18, True, "RND", "WRAP"), A.Fmt(46, 19), A.Fmt(12, 10, True, "RND", "WRAP")])
def test_lossy_wrapping_accumulators_on_16_bit_types(n_taps, ftype, fa):
    """Class B: every product is quantised into ACC_TYPE on its own (AC_TRN / AC_RND), the accumulator wraps -- the sum has no
    order.  16-bit samples and coefficients run fir_lossy_kernel (8 outputs per lane, int32 products); shifts of 1 and 2 bits (64-bit
    accumulation per tap), 12 bits, 28 + bits (everything rounds away but the largest products), an unsigned and a very narrow accumulator;
    history across calls, a ragged last tile, per-channel coefficient sets.  Against the oracle's MAC loops in the reference's order."""
    fin, fc = A.Fmt(16, 2), A.Fmt(16, 2)
    fo = A.Fmt(16, 2, True, "RND", "SAT") if fa.S else A.Fmt(22, 5, True, "RND", "SAT")   # (signed containers: torch has no uint16)
    n = 2 * 4096 + 600 + n_taps
    # (round 5: up to 15 dropped bits, a signed accumulator and <= ~130 taps of a shared set go to the matrix-core class-B kernel first -- whole
    # 4096-sample chunks there, the rest on the exact-order kernel; everything else keeps fir_lossy_kernel.  Which one: tests/test_pathmap_gpu.py)
    fir = check_case(n_taps, ftype, fin, fc, fa, fo, n_ch=3, n=n, splits=[5, 4096 + 5], seed=n_taps + fa.W)
    assert fir.kernel in ("lossy16", "mfma_lossy"), fir.kernel
    if ftype == "SHIFT_REG" and fa.F < 11 + 14:       # narrower sample / unsigned coefficient types: 25 fraction bits in a product
        check_case(n_taps, ftype, A.Fmt(12, 1), A.Fmt(15, 1, False), fa, fo, n_ch=4, n=n, per_channel=True, splits=[2047], seed=n_taps + fa.F,
                   expect_path="generic")


@pytest.mark.parametrize("n_taps", [1, 31, 127, 255, 300])
@pytest.mark.parametrize("ftype", ["SHIFT_REG", "C_BUFF", "FOLD_EVEN", "TRANSPOSED"])
def test_unsigned_16_bit_samples_run_the_matrix_cores(n_taps, ftype):
    """Offset-binary samples: <16,I,false> has 17 significant bits as a signed number.  The engine hands the int8 kernel a sign-flipped image of
    the rows and of the history and adds 32768 * sum(c) to the correction constant; the state stays raw (blobs, reset, fallbacks).  Full-scale
    samples, history across calls, a state blob carried into a second handle, per-channel sets; narrower unsigned types need no flip."""
    kind = "const" if ftype == "TRANSPOSED" else "load"
    fin, fc, fa = A.Fmt(16, 2, False), A.Fmt(16, 2), A.Fmt(44, 16)
    nt = n_taps + (n_taps % 2 if ftype == "FOLD_EVEN" else 0)
    from bench import windowed_sinc_raw
    for k, fo in enumerate((A.Fmt(16, 3, True, "RND", "SAT"), A.Fmt(44, 16), A.Fmt(16, 3, True, "TRN", "WRAP"))):
        cs = windowed_sinc_raw(max(nt | 1, 3), 0.1, fc.F)[:nt] >> 2 if k == 2 else np.minimum(rand_raw(np.random.default_rng(nt + k), fc, (nt,)), 32639)
        check_case(nt, ftype, fin, fc, fa, fo, n_ch=4, n=2048 + 2 * nt + 24, kind=kind, splits=[5, 1100], seed=nt + fo.W, expect_path="mfma_i8",
                   coeffs=np.asarray(cs, dtype=np.int64))
    # state blob: raw samples
    rng = np.random.default_rng(nt)
    c = np.minimum(rand_raw(rng, fc, (nt,)), 32639)
    a1 = A.Fir(nt, ftype, fin, fc, fa, A.Fmt(16, 3, True, "RND", "SAT"), n_channels=3, kind=kind)
    a2 = A.Fir(nt, ftype, fin, fc, fa, A.Fmt(16, 3, True, "RND", "SAT"), n_channels=3, kind=kind)
    a1.set_coeffs(c); a2.set_coeffs(c)
    x = rand_raw(rng, fin, (3, 700 + nt))
    x[0, :] = 65535
    run_engine(a1, x[:, :333])
    a2.set_state(a1.state())
    assert np.array_equal(run_engine(a1, x[:, 333:]), run_engine(a2, x[:, 333:]))
    orc = OracleFir(nt, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(A.Fmt(16, 3, True, "RND", "SAT")), n_ch=3)
    yo = orc.run(c, x)
    a1.reset()
    assert np.array_equal(run_engine(a1, x, [333]), yo)
    if ftype == "SHIFT_REG" and nt <= 257:             # (a dense set per channel beyond 9 K-blocks has no register-resident kernel)
        check_case(nt, ftype, A.Fmt(15, 2, False), fc, fa, A.Fmt(16, 3, True, "RND", "SAT"), n_ch=4, n=1500, per_channel=True,
                   coeffs=np.minimum(rand_raw(rng, fc, (4, nt)), 32639), splits=[777], seed=nt, expect_path="mfma_i8")
        check_case(nt, ftype, fin, fc, fa, A.Fmt(16, 3, True, "RND", "SAT"), n_ch=4, n=1500, per_channel=True,
                   coeffs=np.minimum(rand_raw(rng, fc, (4, nt)), 32639), splits=[777], seed=nt + 1, expect_path="mfma_i8")


LOSSY_MFMA_TYPES = [
    # (IN, COEFF, ACC): s = F_in + F_c - F_acc dropped bits per tap
    (A.Fmt(28, 6), A.Fmt(23, 7), A.Fmt(64, 32)),                      # the reference's prog testbench (rtest_ac_fir_prog_coeffs.cpp:47-54): s = 6
    (A.Fmt(28, 6), A.Fmt(23, 7), A.Fmt(64, 32, True, "RND", "WRAP")),
    (A.Fmt(32, 16), A.Fmt(24, 8), A.Fmt(56, 28, True, "RND", "WRAP")),  # s = 4
    (A.Fmt(32, 16), A.Fmt(24, 8), A.Fmt(40, 16)),                     # s = 8, an accumulator that wraps (40 + 8 <= 64: mod-2^64 sums suffice)
    (A.Fmt(20, 4), A.Fmt(18, 2), A.Fmt(50, 19, True, "RND", "WRAP")),  # s = 1, 20-bit samples in 4-byte containers (all four byte planes)
    (A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(48, 28)),                     # 16-bit samples, wide coefficients: s = 4
    (A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(48, 31, True, "RND", "WRAP")),  # s = 7
    (A.Fmt(15, 3, False), A.Fmt(24, 4), A.Fmt(46, 16)),               # unsigned 15-bit samples: s = 2
    (A.Fmt(28, 6), A.Fmt(23, 7), A.Fmt(60, 34, True, "RND", "WRAP")),  # s = 12: the packed 16-bit sums are emptied every 8 iterations
    (A.Fmt(32, 16), A.Fmt(24, 8), A.Fmt(48, 31)),                     # s = 15: ... every iteration
    (A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(36, 14, True, "RND", "WRAP")),  # 16-bit types, s = 6 (ahead of fir_lossy_kernel since round 5)
]


@pytest.mark.parametrize("ftype", FTYPES6)
@pytest.mark.parametrize("types", range(len(LOSSY_MFMA_TYPES)))
def test_lossy_wrapping_accumulators_on_the_matrix_cores(ftype, types):
    """Class B beyond 16-bit types and through the folds (reference ac_fir_prog_coeffs.h:147-227): sum_k Q(p_k) = (sum_k p_k + N h -
    sum_k ((p_k + h) mod 2^s)) >> s -- the exact sum on the matrix cores (effective taps of the ftype), the dropped bits from the low s bits of
    every (folded) sample and coefficient.  Calls that are whole chunks (the ring kernel), ragged calls (its exact-order tail), history across
    calls, odd and even tap counts (FOLD_EVEN drops the centre tap of an odd count, FOLD_ODD ignores reg[N/2] of an even one), OUT = ACC and
    narrower saturating OUT_TYPEs.  TRANSPOSED keeps reg_trans outside the const class: those handles stay on the exact-order kernel."""
    fin, fc, fa = LOSSY_MFMA_TYPES[types]
    chunk = 4096 if fin.W <= 16 else 2048
    kind = "const" if ftype == "TRANSPOSED" else "prog"
    for k, (n_taps, fo) in enumerate(((27, fa), (28, A.Fmt(32, 12, True, "RND", "SAT")), (97, A.Fmt(fa.W, fa.I)), (6, A.Fmt(30, 9, True, "TRN", "WRAP")))):
        if fin.W <= 16 and k == 1:
            fo = A.Fmt(16, 6, True, "RND", "SAT")
        rng = np.random.default_rng(100 * types + k)
        c = rand_raw(rng, fc, (n_taps,)) >> (max(fc.W - 22, 0) + (2 if n_taps > 30 else 0))    # three balanced base-256 digits per tap
        # FOLD_ODD keeps the pre-add in ACC_TYPE: an accumulator without the extra integer bit lets it wrap -- exact-order kernel
        want = "generic" if (ftype == "FOLD_ODD" and fa.I < fin.I + 1 + (0 if fin.S else 1)) else "mfma_lossy"
        fir = check_case(n_taps, ftype, fin, fc, fa, fo, n_ch=3, n=4 * chunk + 333, kind=kind, splits=[2 * chunk, 2 * chunk + 77], seed=types + k,
                         coeffs=c, expect_path=want)
        del fir
    if ftype == "TRANSPOSED":
        check_case(27, ftype, fin, fc, fa, fa, n_ch=2, n=chunk + 50, kind="load", splits=[chunk], seed=types, expect_path="generic")


def test_lossy_matrix_core_class_bounds():
    """What the class refuses: more than 15 dropped bits, a 64-bit sum that could leave int64 while ACC_TYPE keeps its top bits, sign-dependent
    rounding, a saturating accumulator that could saturate, a coefficient set per channel -- all of them bit-exact on the other kernels."""
    x28, c23 = A.Fmt(28, 6), A.Fmt(23, 7)
    for fa, fc, want in ((A.Fmt(64, 42), c23, "generic"),                               # s = 16
                         (A.Fmt(64, 32, True, "TRN_ZERO", "WRAP"), c23, "generic"),
                         (A.Fmt(64, 32, True, "TRN", "SAT"), c23, "mfma_lossy"),          # saturating, but 27 * 2^22 * 2^27 / 2^6 is far inside 63 bits: never fires
                         (A.Fmt(46, 14, True, "TRN", "SAT"), c23, "generic"),             # ... 45 bits are not enough for this set: exact order
                         (A.Fmt(60, 28, False, "TRN", "WRAP"), c23, "generic"),
                         (A.Fmt(64, 32), A.Fmt(40, 7), "generic")):                     # 2^27 * 27 * 2^39 passes 2^62 and ACC keeps all 64 bits
        check_case(27, "FOLD_ODD", x28, fc, fa, A.Fmt(fa.W, fa.I, fa.S), n_ch=2, n=2048 + 100, kind="prog", splits=[2048], seed=fa.I, expect_path=want)
    check_case(27, "SHIFT_REG", x28, c23, A.Fmt(64, 32), A.Fmt(64, 32), n_ch=3, n=2048 + 100, kind="prog", per_channel=True, splits=[2048], expect_path="generic")


@pytest.mark.parametrize("acc_w", [27, 28, 29, 30])
def test_unsigned_16_bit_samples_against_an_accumulator_inside_twice_the_signed_bound(acc_w):
    """The flipped image is signed (|x| <= 2^15) but the recombined sum is the unsigned dot product, up to 65535 * sum|c|: an ACC_TYPE that holds
    32768 * sum|c| and not 65535 * sum|c| wraps in the reference and must wrap here (round-4 advisor finding: the no-wrap proof of the fast
    epilogues used the signed bound).  32 taps of +100, full-scale samples: <28,0> wraps 6399 -> -1792."""
    fin, fc = A.Fmt(16, 2, False), A.Fmt(16, 2)
    c = np.full(32, 100, dtype=np.int64)
    for fo in (A.Fmt(16, 3, True, "RND", "SAT"), A.Fmt(16, 3, True, "TRN", "WRAP"), A.Fmt(acc_w, acc_w - 28), A.Fmt(24, 4, True, "RND", "SAT")):   # ACC keeps all 28 fraction bits
        rng = np.random.default_rng(acc_w)
        x = rand_raw(rng, fin, (3, 1500))
        x[0, :] = 65535
        x[1, ::2] = 65535
        fir = A.Fir(32, "SHIFT_REG", fin, fc, A.Fmt(acc_w, acc_w - 28), fo, n_channels=3)
        fir.set_coeffs(c)
        assert fir.path == "mfma_i8", fir.path
        y = run_engine(fir, x, [700])
        yo = OracleFir(32, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(A.Fmt(acc_w, acc_w - 28)), ofmt(fo), n_ch=3).run(c, x)
        assert np.array_equal(y, yo), (acc_w, fo.W, y[0, 40:44], yo[0, 40:44])


@pytest.mark.parametrize("fo", [A.Fmt(12, 1, True, "RND", "SAT"), A.Fmt(14, 2, True, "TRN", "WRAP"), A.Fmt(24, 6, True, "RND", "SAT"), A.Fmt(32, 12)])
def test_narrow_and_4_byte_outputs_with_a_coefficient_set_per_channel(fo):
    """The NAR / W4 instantiations read per-channel Toeplitz fragments like the 16-bit kernels do."""
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(42, 14)
    for n_taps in (31, 200):
        c = np.minimum(rand_raw(np.random.default_rng(n_taps + fo.W), fc, (7, n_taps)), 32639) >> 3
        check_case(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_ch=7, n=4096 + n_taps, per_channel=True, coeffs=c, splits=[1024 + 8], seed=fo.W,
                   expect_path="mfma_i8")


@pytest.mark.parametrize("n_taps", [1, 7, 8, 9, 63, 255, 300])
@pytest.mark.parametrize("ftype", ["SHIFT_REG", "ROTATE_SHIFT", "C_BUFF"])
@pytest.mark.parametrize("fa", [A.Fmt(30, 4, True, "TRN", "SAT"), A.Fmt(24, 8, True, "RND", "SAT"), A.Fmt(32, 6, True, "TRN", "SAT"), A.Fmt(30, 2, True, "TRN", "SAT"),
                                A.Fmt(20, 3, False, "TRN", "SAT"), A.Fmt(14, 3, True, "RND", "SAT")])
def test_saturating_accumulators_on_16_bit_types_keep_the_reference_tap_order(n_taps, ftype, fa):
    """Class C: acc = sat(acc + Q(p)) after every tap; the clamp makes the order matter -- descending taps for SHIFT_REG / ROTATE_SHIFT,
    ascending for C_BUFF.  Dense full-range coefficients drive the accumulator into both rails and back; shifts of 0, 2 and 12 + bits, an
    unsigned and a very narrow accumulator; history across calls, a ragged last tile, per-channel sets.  Against the oracle's MAC loops."""
    fin, fc = A.Fmt(16, 2), A.Fmt(16, 2)
    fo = A.Fmt(16, 2, True, "RND", "SAT") if fa.S else A.Fmt(22, 5, True, "RND", "SAT")
    n = 2048 + 600 + n_taps

    def sat_free(c):   # engine_fir.hip: acdsp_fir::sat_free -- the set cannot drive a signed accumulator to its bounds (then it is a wrapping one)
        if not fa.S:
            return False
        s = fin.F + fc.F - fa.F
        worst = max(int(np.abs(row).sum()) for row in np.atleast_2d(c)) << 15
        worst = (worst << -s) if s <= 0 else (worst >> s) + n_taps + 1
        return worst <= (1 << (fa.W - 1)) - 1

    for n_ch, per_channel, splits, seed in ((3, False, [5, 1200], n_taps + fa.W), (4, True, [2047], n_taps + fa.F)):
        c = rand_raw(np.random.default_rng(1000 + seed), fc, (n_ch, n_taps) if per_channel else (n_taps,))
        fir = check_case(n_taps, ftype, fin, fc, fa, fo, n_ch=n_ch, n=n, per_channel=per_channel, splits=splits, seed=seed, coeffs=c)
        assert sat_free(c) or fir.path == "generic", fir.path      # (a set that cannot saturate may still land on the exact-order family: one tap, per-channel sets)


@pytest.mark.parametrize("n_taps", [5, 29, 64, 127])
@pytest.mark.parametrize("fin,fc", [(A.Fmt(16, 8), A.Fmt(32, 16)), (A.Fmt(16, 2), A.Fmt(24, 4)), (A.Fmt(32, 16), A.Fmt(16, 2)), (A.Fmt(32, 16), A.Fmt(24, 8))])
@pytest.mark.parametrize("fo", [A.Fmt(56, 26), A.Fmt(32, 12, True, "RND", "SAT"), A.Fmt(16, 4, True, "RND", "SAT"), None])
def test_plain_firs_with_wide_samples_or_coefficients_on_the_ring_kernel(n_taps, fin, fc, fo):
    """The reference testbench's own type family (16-bit samples x 32-bit coefficients) and 32-bit samples: multi-plane MFMA FIR without decimation
    (fir_gen_ring_kernel at R = 1 -- on int16 one 1 KB load feeds two 256-output steps), whole chunks on it, the ragged tail and the call edges on the
    general kernel.  Band-limited sets (two / three coefficient digits), history across calls."""
    from bench import windowed_sinc_raw
    fa = A.Fmt(60, 60 - fin.F - fc.F)
    if fo is None:                                        # the testbench's own ACC = OUT <64,32> (rtest_ac_fir_const_coeffs.cpp:71-74)
        fa = fo = A.Fmt(64, 32)
        if fin.F + fc.F > 32:
            pytest.skip("<64,32> loses product bits for these types: per-tap class")
    c = np.asarray(windowed_sinc_raw(n_taps | 1, 0.12, fc.F)[:n_taps], dtype=np.int64)
    check_case(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_ch=3, n=3 * 4096 + 700 + n_taps, splits=[4096 + 48], seed=n_taps + fo.W, coeffs=c, expect_path="mfma_gen")
    check_case(n_taps + (n_taps % 2), "FOLD_EVEN", fin, fc, fa, fo, n_ch=2, n=2 * 4096 + 16, seed=n_taps, expect_path="mfma_gen",
               coeffs=np.asarray(windowed_sinc_raw((n_taps + n_taps % 2) | 1, 0.12, fc.F)[:n_taps + n_taps % 2], dtype=np.int64))


@pytest.mark.parametrize("n_taps", [1, 2, 31, 32, 33, 63, 64, 65, 127, 255, 257])
def test_mfma_path_tap_counts(n_taps):
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(42, 14)
    # raw coefficients >= 32640 cannot be split into two signed bytes (covered by the fallback test below)
    c = np.minimum(rand_raw(np.random.default_rng(1000 + n_taps), fc, (n_taps,)), 32639)
    check_case(n_taps, "SHIFT_REG", fin, fc, fa, A.Fmt(16, 2, True, "RND", "SAT"), n_ch=37, n=1000 + n_taps,
               splits=[77, 512], expect_path="mfma_i8", seed=n_taps, coeffs=c)


@pytest.mark.parametrize("n_taps,big", [
    (255, (160, 64)),          # high-byte blocks 3 and 6 of 9: skip 3 low / 2 high
    (255, (192, 96)),          # blocks 2 and 5: skip 2 / 3
    (255, (160, 96)),          # blocks 3 .. 5: skip 3 / 3
    (255, (192, 64)),          # blocks 2 .. 6: skip 2 / 2
    (255, (224, 127)),         # block 1: no skip on the low side -> dense kernel
    (255, ()),                 # no high bytes at all
    (193, (128, 64)),          # NB = 7: blocks 2 and 4: skip 2 / 2
    (193, (96,)),              # NB = 7: block 3 only: skip 3 / 3
    (130, (64,)),              # NB = 6
])
@pytest.mark.parametrize("fo", [A.Fmt(16, 2, True, "RND", "SAT"), A.Fmt(40, 12)])
def test_mfma_high_byte_band_variants(n_taps, big, fo):
    """Coefficient sets whose high-byte Toeplitz planes are non-zero in chosen K-blocks only: every instantiated band skip
    (asymmetric ones included) against the oracle, interior (pipelined) and edge chunks."""
    rng = np.random.default_rng(n_taps + len(big))
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12)
    c = rng.integers(-100, 101, size=n_taps, dtype=np.int64)
    for t in big:
        c[t] = int(rng.integers(3000, 20000)) * (1 if rng.integers(2) else -1)
    check_case(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_ch=5, n=1024 * 70 + 333, kind="load", coeffs=c, expect_path="mfma_i8",
               splits=[1024 * 33 + 5], seed=7)


@pytest.mark.parametrize("ftype", FTYPES6)
def test_mfma_path_ftypes_and_wide_output(ftype):
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12)
    c = windowed_sinc(63, 0.11, fc)
    check_case(63, ftype, fin, fc, fa, fa, n_ch=33, n=900, kind="const", coeffs=c, expect_path="mfma_i8", splits=[64])
    check_case(63, ftype, fin, fc, fa, A.Fmt(16, 2, True, "TRN", "WRAP"), n_ch=64, n=900, kind="const", coeffs=c,
               expect_path="mfma_i8")


@pytest.mark.parametrize("fo", [
    A.Fmt(16, 2, True, "RND", "SAT"),    # rs = 14: the config-2 epilogue
    A.Fmt(16, 2, True, "TRN", "WRAP"),
    A.Fmt(16, 8, True, "RND", "SAT"),    # rs = 20 (> 16): the other shift class of the 32-bit epilogue
    A.Fmt(16, 8, True, "TRN", "WRAP"),
    A.Fmt(16, 1, True, "RND", "WRAP"),   # rs = 13, output wraps
    A.Fmt(16, -8, True, "RND", "SAT"),   # rs = 4: every output saturates or nearly so
    A.Fmt(40, 12),                       # OUT = ACC: 64-bit shift-and-wrap epilogue (config 2, wide row)
    A.Fmt(40, 12, True, "RND", "WRAP"),
    A.Fmt(34, 4, True, "TRN", "WRAP"),   # rs = -2 (left shift) and a wrapping 34-bit result
    A.Fmt(24, 10, True, "RND", "WRAP"),  # rs = 14 into an int32 container: generic epilogue
])
@pytest.mark.parametrize("n_taps", [255, 130, 33])
def test_mfma_pipelined_interior_chunks(fo, n_taps):
    # runs long enough (>= 2 steps of 1024 per chunk) for the software-pipelined body, ragged tail through the
    # edge body, a second call continuing from carried state
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12)
    c = windowed_sinc(n_taps, 0.1, fc)
    check_case(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_ch=5, n=9 * 1024 + 77, coeffs=c, expect_path="mfma_i8",
               splits=[4096], seed=n_taps)


def test_mfma_pipelined_dense_and_per_channel_sets():
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12)
    rng = np.random.default_rng(77)
    c = rng.integers(-32768, 32640, size=255)
    for fo in (A.Fmt(16, 2, True, "RND", "SAT"), fa):
        check_case(255, "SHIFT_REG", fin, fc, fa, fo, n_ch=3, n=8 * 1024, coeffs=c, expect_path="mfma_i8")
    cc = rng.integers(-3000, 3000, size=(4, 200))
    check_case(200, "C_BUFF", fin, fc, fa, A.Fmt(16, 2, True, "RND", "SAT"), n_ch=4, n=6 * 1024 + 5, coeffs=cc,
               per_channel=True, expect_path="mfma_i8", splits=[3000])


@pytest.mark.parametrize("pattern", ["small_only", "big_multiples_of_256", "single_tap", "all_zero", "two_islands"])
def test_mfma_zero_block_skipping(pattern):
    # Toeplitz blocks whose hi or lo byte plane is entirely zero are skipped: exercise every mask shape
    N = 255
    rng = np.random.default_rng(3)
    c = np.zeros(N, dtype=np.int64)
    if pattern == "small_only":
        c[:] = rng.integers(-128, 128, size=N)            # hi plane all zero
    elif pattern == "big_multiples_of_256":
        c[:] = 256 * rng.integers(-100, 100, size=N)      # lo plane all zero
    elif pattern == "single_tap":
        c[200] = -12345
    elif pattern == "two_islands":
        c[10:20] = rng.integers(-30000, 30000, size=10)
        c[230:240] = rng.integers(-100, 100, size=10)
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12)
    check_case(N, "SHIFT_REG", fin, fc, fa, A.Fmt(16, 2, True, "RND", "SAT"), n_ch=3, n=2500, coeffs=c, expect_path="mfma_i8",
               splits=[1024])
    check_case(N, "SHIFT_REG", fin, fc, fa, fa, n_ch=2, n=1100, coeffs=c, expect_path="mfma_i8")


def test_mfma_extreme_values_and_fallback():
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
    N, n_ch, n = 255, 32, 2048
    c = np.full(N, -32768, dtype=np.int64)
    x = np.full((n_ch, n), -32768, dtype=np.int64)
    x[1::2] = 32767
    fir = A.Fir(N, "SHIFT_REG", fin, fc, fa, fo, n_channels=n_ch)
    fir.set_coeffs(c)
    assert fir.path == "mfma_i8"
    orc = OracleFir(N, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    assert np.array_equal(run_engine(fir, x), orc.run(c, x))
    # +32767 cannot be written as two signed bytes -> the engine must leave the int16 MFMA kernel (the
    # generalised kernel takes it with a third coefficient digit), not mis-compute
    c2 = np.full(N, 32767, dtype=np.int64)
    fir.reset()
    fir.set_coeffs(c2)
    assert fir.path == "mfma_gen"
    orc = OracleFir(N, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    assert np.array_equal(run_engine(fir, x[:, :300]), orc.run(c2, x[:, :300]))


def test_generic_equals_fast_paths():
    # same configuration through all three kernel families
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
    for force in (False, True):
        check_case(95, "FOLD_ODD", fin, fc, fa, fo, n_ch=40, n=640, force_generic=force,
                   coeffs=windowed_sinc(95, 0.2, fc), expect_path="generic" if force else "mfma_i8")


def test_reference_fir_testbench_vectors():
    # the three shipped FIR tests (SQNR >= 60 dB against the MATLAB double reference)
    cases = (("const", 29, A.Fmt(16, 8), A.Fmt(32, 16), 84.24), ("load", 27, A.Fmt(32, 16), A.Fmt(32, 16), 89.56),
             ("prog", 27, A.Fmt(28, 6), A.Fmt(23, 7), 89.56))
    for name, taps, fi, fc, want in cases:
        fa = A.Fmt(64, 32)
        c = np.array([from_double(float(v), ofmt(fc)) for v in read_fracs("ac_fir_%s_coeffs_cfg.txt" % name)], dtype=np.int64)
        x = two_tone(ofmt(fi))[None, :]
        fir = A.Fir(taps, "FOLD_ODD", fi, fc, fa, fa, kind=name)
        fir.set_coeffs(c)
        y = fir.run_host(x)[0].astype(np.int64)
        got = sqnr_db(y, 32, read_fracs("ac_fir_%s_coeffs_ref.txt" % name)[:1024])
        assert got >= 60.0 and abs(got - want) < 0.01, (name, got)
        assert np.array_equal(y, OracleFir(taps, "FOLD_ODD", ofmt(fi), ofmt(fc), ofmt(fa), ofmt(fa)).run(c, x)[0])


@pytest.mark.parametrize("n_taps", [63, 255, 300, 1023])
def test_one_sample_host_calls_carry_the_history(n_taps):
    """ac_fir_prog_coeffs::run is one sample per call (reference ac_fir_prog_coeffs.h:281): small host-side calls go through the pinned
    buffers and the register-resident MFMA kernels write the next history themselves (one launch per call, every shape)."""
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(44, 16), A.Fmt(16, 2, True, "RND", "SAT")
    c = windowed_sinc(n_taps, 0.05, fc)
    rng = np.random.default_rng(n_taps)
    x = rng.integers(-32768, 32768, size=(1, 3 * n_taps + 70), dtype=np.int64)
    fir = A.Fir(n_taps, "SHIFT_REG", fin, fc, fa, fo, kind="prog")
    fir.set_coeffs(c)
    assert fir.path == "mfma_i8"
    cuts = [0, 1, 2, 3, 20, 21, n_taps, n_taps + 1, 2 * n_taps + 5, x.shape[1]]
    y = np.concatenate([fir.run_host(x[:, a:b]).astype(np.int64) for a, b in zip(cuts[:-1], cuts[1:])] , axis=1)
    want = OracleFir(n_taps, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo)).run(c, x)
    assert np.array_equal(y, want)


def test_ragged_and_empty_inputs():
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
    fir = check_case(40, "SHIFT_REG", fin, fc, fa, fo, n_ch=1, n=1)            # single sample
    check_case(40, "SHIFT_REG", fin, fc, fa, fo, n_ch=65, n=33, splits=[0, 1])  # odd channel count, tiny calls
    import ctypes as C
    assert A.lib.acdsp_fir_run(fir._h, None, 0, 0, None, 0, None) == 0           # n = 0 is a no-op


def test_unaligned_rows_take_a_correct_path():
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
    rng = np.random.default_rng(11)
    n_ch, n, N = 4, 333, 50
    c = rand_raw(rng, fc, (N,))
    x = rand_raw(rng, fin, (n_ch, n))
    fir = A.Fir(N, "SHIFT_REG", fin, fc, fa, fo, n_channels=n_ch)
    fir.set_coeffs(c)
    big = torch.zeros((n_ch, n + 5), dtype=torch.int16, device="cuda")
    big[:, 3:3 + n] = torch.from_numpy(x).to(torch.int16).cuda()
    y = fir.run(big[:, 3:3 + n]).cpu().numpy().astype(np.int64)   # row start not 16-byte aligned, odd stride
    assert np.array_equal(y, OracleFir(N, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch).run(c, x))


def test_unaligned_long_rows_are_staged_for_the_matrix_core_kernels():
    # odd row start and stride on a long run: the engine copies the rows into an aligned image on the device and still runs
    # the (pipelined) MFMA kernel; same for the wide-input gen kernel; state carries across an aligned / unaligned mix
    rng = np.random.default_rng(12)
    for fin, fc, fa, fo, N, dt, path in ((A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT"), 200, torch.int16, "mfma_i8"),
                                         (A.Fmt(36, 21), A.Fmt(16, 1), A.Fmt(60, 30), A.Fmt(24, 9, True, "RND", "SAT"), 127, torch.int64, "mfma_gen")):
        n_ch, n = 3, 7001
        c = windowed_sinc(N, 0.1, fc)
        x = rand_raw(rng, fin, (n_ch, 2 * n))
        fir = A.Fir(N, "SHIFT_REG", fin, fc, fa, fo, n_channels=n_ch, kind="const")
        fir.set_coeffs(c)
        assert fir.path == path
        big = torch.zeros((n_ch, 2 * n + 7), dtype=dt, device="cuda")
        big[:, 3:3 + 2 * n] = torch.from_numpy(x).to(dt).cuda()
        y1 = fir.run(big[:, 3:3 + n]).cpu().numpy().astype(np.int64)
        y2 = fir.run(big[:, 3 + n:3 + 2 * n]).cpu().numpy().astype(np.int64)
        yo = OracleFir(N, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch).run(c, x)
        assert np.array_equal(np.concatenate([y1, y2], axis=1), yo)


@pytest.mark.parametrize("in_w,out_w", [(1024, 1024), (1024, 1023), (1023, 1024), (1023, 1023)])
def test_ragged_tail_and_unaligned_output_saturate(in_w, out_w):
    # n = 1023 with full-scale random coefficients: the last, incomplete group of four outputs and every output of an
    # unaligned output row leave through the element-wise stores, which must clamp like the packed ones (AC_SAT)
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
    rng = np.random.default_rng(1)
    n_ch, N, n = 9, 200, 1023
    c = np.minimum(rng.integers(-32768, 32768, size=(N,)), 32639)
    x = rng.integers(-32768, 32768, size=(n_ch, n))
    yo = OracleFir(N, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch).run(c, x)
    assert (np.abs(yo) >= 32767).mean() > 0.3          # most outputs saturate
    fir = A.Fir(N, "SHIFT_REG", fin, fc, fa, fo, n_channels=n_ch)
    fir.set_coeffs(c)
    assert fir.path == "mfma_i8"
    xin = torch.zeros((n_ch, in_w), dtype=torch.int16, device="cuda")
    xin[:, :n] = torch.from_numpy(x).to(torch.int16).cuda()
    out = torch.zeros((n_ch, out_w), dtype=torch.int16, device="cuda")
    y = fir.run(xin[:, :n], out)[:, :n].cpu().numpy().astype(np.int64)
    assert np.array_equal(y, yo)


def test_anti_ftypes_are_rejected():
    with pytest.raises(A.AcdspError):
        A.Fir(8, "FOLD_EVEN_ANTI", A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2))


def test_device_stimulus_matches_oracle():
    t = torch.empty((5, 1000), dtype=torch.int16, device="cuda")
    A.fill_stimulus(t, 0xACD5, 16, ch0=3, t0=17)
    assert np.array_equal(t.cpu().numpy().astype(np.int64), stimulus(0xACD5, 5, 1000, 16, ch0=3, t0=17))
    t = torch.empty((2, 77), dtype=torch.int32, device="cuda")
    A.fill_stimulus(t, 7, 32)
    assert np.array_equal(t.cpu().numpy().astype(np.int64), stimulus(7, 2, 77, 32))


def test_config2_full_size_properties():
    """BASELINE config 2 at full size: 255 taps, 1024 channels x 2^20 samples, <16,2> in/out.
    Checked by (a) oracle on sampled channels/windows and (b) linearity y(x1+x2) = y(x1)+y(x2) in the
    wide (lossless) output, a size-independent property of the exact integer FIR."""
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12)
    fo = A.Fmt(16, 2, True, "RND", "SAT")
    N, n_ch, n = 255, 1024, 1 << 20
    c = windowed_sinc(N, 0.1, fc)
    x = torch.empty((n_ch, n), dtype=torch.int16, device="cuda")
    A.fill_stimulus(x, 0xACD5, 16)
    fir = A.Fir(N, "SHIFT_REG", fin, fc, fa, fo, n_channels=n_ch)
    fir.set_coeffs(c)
    assert fir.path == "mfma_i8"
    y = fir.run(x)
    torch.cuda.synchronize()
    for ch, t0 in ((0, 0), (1, 5000), (517, 600000), (1023, n - 4096)):
        w = 4096
        xs = stimulus(0xACD5, 1, w + N - 1, 16, ch0=ch, t0=max(t0 - (N - 1), 0))
        if t0 == 0:
            xs = np.concatenate([np.zeros((1, N - 1), dtype=np.int64), xs[:, :w]], axis=1)
        yo = OracleFir(N, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo)).run(c, xs)[0][N - 1:]
        assert np.array_equal(y[ch, t0:t0 + w].cpu().numpy().astype(np.int64), yo), (ch, t0)
    # linearity on a 64-channel slice with the wide output
    firw = A.Fir(N, "SHIFT_REG", fin, fc, fa, fa, n_channels=64)
    firw.set_coeffs(c)
    x1 = (x[:64] >> 1)
    x2 = (x[64:128] >> 1)
    y1 = firw.run(x1); firw.reset()
    y2 = firw.run(x2); firw.reset()
    y12 = firw.run(x1 + x2)
    assert torch.equal(y12, y1 + y2)


@pytest.mark.parametrize("n_taps", [258, 300, 511, 800, 1023, 1025])
def test_mfma_path_large_tap_counts(n_taps):
    # more than 257 taps: Toeplitz fragments live in LDS (shared coefficient set)
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(44, 16)
    c = np.minimum(rand_raw(np.random.default_rng(n_taps), fc, (n_taps,)), 32639)
    check_case(n_taps, "SHIFT_REG", fin, fc, fa, A.Fmt(16, 2, True, "RND", "SAT"), n_ch=11, n=2500 + n_taps, coeffs=c,
               expect_path="mfma_i8", splits=[1024, 1500], seed=n_taps)
    check_case(n_taps, "C_BUFF", fin, fc, fa, fa, n_ch=3, n=1100 + n_taps, coeffs=windowed_sinc(n_taps, 0.07, fc),
               expect_path="mfma_i8")


@pytest.mark.parametrize("n_taps,fo", [(1023, A.Fmt(16, 2, True, "RND", "SAT")), (700, A.Fmt(16, 2, True, "TRN", "WRAP")),
                                       (300, A.Fmt(16, 8, True, "RND", "SAT"))])
def test_mfma_large_tap_counts_double_wide_chunks(n_taps, fo):
    # long enough for complete chunks of 2048-output steps (fir_mfma_big2_kernel) + a ragged rest on the single-wide
    # kernel; 9 channels = one full workgroup of 8 waves + one with surplus waves
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(44, 16)
    c = windowed_sinc(n_taps, 0.05, fc)
    check_case(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_ch=9, n=2 * 8192 + 1500, coeffs=c, expect_path="mfma_i8", splits=[8192 + 8],
               seed=n_taps)


@pytest.mark.parametrize("n_taps", [258, 289, 290, 321, 352, 383, 416, 449, 480, 511, 513, 514, 546, 577, 640, 705, 769, 830, 897, 930, 961, 975, 992])
@pytest.mark.parametrize("fo", [A.Fmt(16, 2, True, "RND", "SAT"), A.Fmt(16, 2, True, "TRN", "WRAP")])
def test_mfma_mid_tap_counts_register_resident_shapes(n_taps, fo):
    """258 - 961 taps (10 .. 31 K-blocks): unit-gain low-pass sets run on the register-resident shapes of fir_mfma_mid*.hip (an even
    plan is padded by one zero block: the history reaches one block further back), every other set on the LDS-resident kernels."""
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(44, 16)
    c = windowed_sinc(n_taps, 0.08, fc)
    fir = check_case(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_ch=5, n=3 * 8192 + 700 + n_taps, coeffs=c, expect_path="mfma_i8",
                     splits=[1, 8192 + 24, 9000], seed=n_taps)
    nb = (n_taps - 1 + 31) // 32 + 1
    nb += (nb % 2 == 0)
    assert fir.mfma_issued() == 2 * nb + 2 * 5, (fir.mfma_issued(), nb)      # five high-byte blocks: the mid shapes took the set
    # a set with large coefficients at both ends has no central band: LDS-resident kernel, same answers
    c2 = c.copy()
    c2[0] = c2[-1] = 9000
    fir2 = check_case(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_ch=3, n=8192 + 300, coeffs=c2, expect_path="mfma_i8", splits=[4100], seed=n_taps + 1)
    assert fir2.mfma_issued() > 2 * nb + 2 * 5


@pytest.mark.parametrize("n_taps", [511, 767, 1023])
@pytest.mark.parametrize("fo", [A.Fmt(16, 2, True, "RND", "SAT"), A.Fmt(16, 2, True, "TRN", "WRAP"), A.Fmt(12, 0, True, "RND", "WRAP")])
@pytest.mark.parametrize("fa", [A.Fmt(44, 16), A.Fmt(38, 10)])
def test_dense_sets_past_the_32_bit_epilogue_bounds(n_taps, fo, fa):
    """DENSE random 16-bit sets of 511+ taps: 2^8 mid + ll no longer fits int32, and with ACC <38,10> the gain of 2^9 wraps the
    accumulator (ac_fir_prog_coeffs.h:147-155: every `acc +=` wraps into ACC_TYPE; for the lossless class = one wrap of the exact
    sum).  Round 3 sent these to the generic epilogue (4.6 - 6.1 ms on the bench shape); the 64-bit branch-free class (EPI 4) keeps
    them on the matrix cores with the packed tile stores.  SAT / WRAP outputs, narrow OUT in a 16-bit container, split calls."""
    fin, fc = A.Fmt(16, 2), A.Fmt(16, 2)
    rng = np.random.default_rng(n_taps)
    c = np.minimum(rand_raw(rng, fc, (n_taps,)), 32639)
    fir = check_case(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_ch=9, n=2 * 8192 + 1500, coeffs=c, expect_path="mfma_i8", splits=[8192 + 8],
                     seed=n_taps + 7)
    nb = (n_taps - 1 + 31) // 32 + 1
    nb += (nb % 2 == 0 and 10 <= nb <= 32)
    assert fir.mfma_issued() == 4 * nb, (fir.mfma_issued(), nb)       # every block of both planes


@pytest.mark.parametrize("n_taps", [300, 640, 1023])
def test_per_channel_coefficients_beyond_257_taps(n_taps):
    """A coefficient set per channel needs the register-resident kernels; beyond 9 K-blocks those exist for band-limited sets
    (every object of a bank of ac_fir_prog_coeffs filters with its own low-pass): matrix-core path.  Dense sets: exact-order path."""
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(44, 16), A.Fmt(16, 2, True, "RND", "SAT")
    n_ch = 7
    c = np.stack([windowed_sinc(n_taps, 0.03 + 0.01 * ch, fc, gain=0.5 + 0.07 * ch) for ch in range(n_ch)])
    fir = check_case(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_ch=n_ch, n=2 * 8192 + 333, per_channel=True, coeffs=c, expect_path="mfma_i8",
                     splits=[8192 + 16, 9001], seed=n_taps)
    assert fir.mfma_issued() > 0
    dense = np.minimum(rand_raw(np.random.default_rng(n_taps), fc, (n_ch, n_taps)), 32639) // 64
    fir2 = check_case(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_ch=n_ch, n=1500, per_channel=True, coeffs=dense, seed=n_taps + 1)
    assert fir2.path != "mfma_i8", fir2.path


def test_config4_shape_1023_taps_prog_coeffs():
    """BASELINE config 4 shape (ac_fir_prog_coeffs, 1023 taps, <16,2>, ACC <42,14>) at a reduced size."""
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(42, 14)
    c = windowed_sinc(1023, 0.05, fc)
    check_case(1023, "SHIFT_REG", fin, fc, fa, A.Fmt(16, 2, True, "RND", "SAT"), n_ch=5, n=3000, kind="prog", coeffs=c,
               splits=[1, 1500], expect_path="mfma_i8")
    check_case(1023, "FOLD_ODD", fin, fc, fa, fa, n_ch=2, n=1200, kind="prog", coeffs=c)


def test_config5_ddc_cascade_cic_then_fir():
    """BASELINE config 5 shape: ac_cic_dec_full R=16 N=5 on <16,1> -> INT <36,21> -> 127-tap
    ac_fir_const_coeffs with IN=<36,21>, COEFF=<16,1>; I and Q are two real streams (the CIC takes plain
    ac_fixed only, ac_cic_dec_full.h:76).  The engine output of the cascade must equal the oracle cascade."""
    from oracle import OracleCic
    cin = A.Fmt(16, 1)
    cic = A.Cic(False, 16, 1, 5, cin, cin, n_channels=6)
    it = cic.int_type
    assert (it.W, it.I) == (36, 21)
    mid = A.Fmt(it.W, it.I)
    fc, fa, fo = A.Fmt(16, 1), A.Fmt(60, 30), A.Fmt(24, 9, True, "RND", "SAT")
    c = windowed_sinc(127, 0.2, fc)
    n_ch, n = 6, 16 * 700            # 3 complex channels = 6 real streams
    x = stimulus(0xDDC, n_ch, n, 16)
    cic = A.Cic(False, 16, 1, 5, cin, mid, n_channels=n_ch)
    fir = A.Fir(127, "SHIFT_REG", mid, fc, fa, fo, n_channels=n_ch, kind="const")
    fir.set_coeffs(c)
    assert fir.path == "mfma_gen"            # 36-bit input: five byte planes on the matrix cores
    oc = OracleCic(0, 16, 1, 5, ofmt(cin), ofmt(mid), n_ch=n_ch)
    of = OracleFir(127, "SHIFT_REG", ofmt(mid), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    for a, b in ((0, 16 * 300), (16 * 300, n)):      # two bursts: both stages carry state
        xd = torch.from_numpy(x[:, a:b].copy()).to(torch.int16).cuda()
        u = cic.run(xd)                               # int64 containers (36-bit words)
        y = fir.run(u.contiguous()).cpu().numpy().astype(np.int64)
        yo = of.run(c, oc.run(x[:, a:b]))
        assert np.array_equal(y, yo)


@pytest.mark.parametrize("fin,fc,n_taps", [(A.Fmt(24, 8), A.Fmt(16, 2), 63), (A.Fmt(32, 16), A.Fmt(18, 2), 100),
                                           (A.Fmt(36, 21), A.Fmt(16, 1), 127), (A.Fmt(48, 20), A.Fmt(12, 2), 33),
                                           (A.Fmt(16, 2), A.Fmt(24, 4), 40), (A.Fmt(20, 4, False), A.Fmt(10, 1), 17)])
def test_generalised_mfma_kernel_wide_types(fin, fc, n_taps):
    """Inputs / coefficients wider than 16 bits: multi-plane int8 MFMA, exact mod 2^64 then ACC wrap."""
    fa = A.Fmt(64, 64 - (fin.F + fc.F))
    rng = np.random.default_rng(n_taps)
    c = rand_raw(rng, fc, (n_taps,))
    for ftype, fo in (("SHIFT_REG", fa), ("FOLD_ODD", A.Fmt(30, 12, True, "RND_CONV", "SAT")), ("C_BUFF", A.Fmt(64, 40))):
        check_case(n_taps, ftype, fin, fc, fa, fo, n_ch=3, n=1200, coeffs=c, expect_path="mfma_gen", splits=[16, 500],
                   seed=n_taps + 1)


def test_scale_edges_long_stream_and_many_channels():
    """One very long channel (2^24 samples, many chunks per channel) and many short channels (8192 x 2048)."""
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12)
    fo = A.Fmt(16, 2, True, "RND", "SAT")
    c = windowed_sinc(255, 0.1, fc)
    # long stream: check three windows against the oracle
    n = 1 << 24
    x = torch.empty((1, n), dtype=torch.int16, device="cuda")
    A.fill_stimulus(x, 5, 16)
    fir = A.Fir(255, "SHIFT_REG", fin, fc, fa, fo, n_channels=1)
    fir.set_coeffs(c)
    y = fir.run(x)
    for t0 in (0, 5_000_000, n - 3000):
        w = 3000
        xs = stimulus(5, 1, w + 254, 16, t0=max(t0 - 254, 0))
        if t0 == 0:
            xs = np.concatenate([np.zeros((1, 254), dtype=np.int64), xs[:, :w]], axis=1)
        yo = OracleFir(255, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo)).run(c, xs)[0][254:]
        assert np.array_equal(y[0, t0:t0 + w].cpu().numpy().astype(np.int64), yo), t0
    # many channels
    n_ch, n2 = 8192, 2048
    x2 = torch.empty((n_ch, n2), dtype=torch.int16, device="cuda")
    A.fill_stimulus(x2, 6, 16)
    fir2 = A.Fir(255, "SHIFT_REG", fin, fc, fa, fo, n_channels=n_ch)
    fir2.set_coeffs(c)
    y2 = fir2.run(x2)
    for ch in (0, 4097, 8191):
        yo = OracleFir(255, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo)).run(c, stimulus(6, 1, n2, 16, ch0=ch))[0]
        assert np.array_equal(y2[ch].cpu().numpy().astype(np.int64), yo), ch


def test_two_handles_on_two_streams_are_independent():
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
    c1, c2 = windowed_sinc(127, 0.1, fc), windowed_sinc(127, 0.3, fc)
    xs = [stimulus(70 + i, 8, 50000, 16) for i in range(2)]
    firs = [A.Fir(127, "SHIFT_REG", fin, fc, fa, fo, n_channels=8) for _ in range(2)]
    firs[0].set_coeffs(c1); firs[1].set_coeffs(c2)
    streams = [torch.cuda.Stream() for _ in range(2)]
    xd = [torch.from_numpy(x).to(torch.int16).cuda() for x in xs]
    torch.cuda.synchronize()
    outs = [None, None]
    for rep in range(3):        # interleaved launches on two streams, state carried per handle
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                o = firs[i].run(xd[i])
                outs[i] = o if rep == 2 else outs[i]
    torch.cuda.synchronize()
    for i, c in enumerate((c1, c2)):
        orc = OracleFir(127, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=8)
        for rep in range(3):
            yo = orc.run(c, xs[i])
        assert np.array_equal(outs[i].cpu().numpy().astype(np.int64), yo)


@pytest.mark.parametrize("fo", [A.Fmt(24, 9, True, "RND", "SAT"), A.Fmt(24, 9, True, "TRN", "WRAP"), A.Fmt(30, 9, True, "RND", "WRAP"),
                                A.Fmt(60, 30), A.Fmt(24, 9, True, "RND_ZERO", "SAT")])
def test_wide_input_long_run_takes_the_branch_free_gen_kernel(fo):
    # DDC stage B: 127 taps on the CIC's 36-bit words (int64 containers, five byte planes); the last OUT_TYPE is
    # outside the fast conversion (general kernel for everything)
    fin, fc, fa = A.Fmt(36, 21), A.Fmt(16, 1), A.Fmt(60, 30)
    c = windowed_sinc(127, 0.2, fc)
    check_case(127, "SHIFT_REG", fin, fc, fa, fo, n_ch=3, n=5000, kind="const", coeffs=c, expect_path="mfma_gen", splits=[2048],
               seed=9)


# ---- ac_fir_reg_share: tap-ordered coefficients, ascending MAC order, anti-symmetric folds ----

RS_FTYPES = ["SHIFT_REG", "FOLD_EVEN", "FOLD_EVEN_ANTI", "FOLD_ODD", "FOLD_ODD_ANTI"]


def check_reg_share(n_taps, ftype, fin, fc, fa, fo, n_ch=3, n=700, splits=None, seed=0, coeffs=None, expect_path=None):
    rng = np.random.default_rng(seed)
    x = rand_raw(rng, fin, (n_ch, n))
    if coeffs is None:
        coeffs = rand_raw(rng, fc, (n_taps,))
    fir = A.Fir(n_taps, ftype, fin, fc, fa, fo, n_channels=n_ch, kind="reg_share")
    fir.set_coeffs(coeffs)
    if expect_path:
        assert fir.path == expect_path, fir.path
    y = run_engine(fir, x, splits)
    yo = OracleFir(n_taps, ftype, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch, reg_share=(1, 1, 0)).run(coeffs, x)
    bad = np.argwhere(y != yo)
    assert bad.size == 0, "%d mismatches, first at %s (path %s)" % (len(bad), bad[0], fir.path)


@pytest.mark.parametrize("ftype", RS_FTYPES)
def test_reg_share_lossless_runs_on_the_matrix_cores(ftype):
    fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12)
    n_taps = 63 if "ODD" in ftype else 64
    c = np.clip(rand_raw(np.random.default_rng(4), fc, (n_taps,)), -16000, 16000)    # +/- sums of two taps stay int8-splittable
    check_reg_share(n_taps, ftype, fin, fc, fa, A.Fmt(16, 2, True, "RND", "SAT"), n_ch=5, n=3 * 1024 + 9, coeffs=c,
                    expect_path="mfma_i8", splits=[1024], seed=1)
    check_reg_share(n_taps, ftype, fin, fc, fa, fa, n_ch=2, n=900, coeffs=c, expect_path="mfma_i8", seed=2)
    # reference usage-example types (ac_fir_reg_share.h:50-53): <32,16> data and coefficients, <64,32> MAC type
    check_reg_share(27 if "ODD" in ftype else 28, ftype, A.Fmt(32, 16), A.Fmt(32, 16), A.Fmt(64, 32), A.Fmt(64, 32), splits=[100, 113],
                    seed=3)


@pytest.mark.parametrize("ftype", RS_FTYPES)
@pytest.mark.parametrize("q,o", [("TRN", "WRAP"), ("RND_CONV", "SAT"), ("TRN_ZERO", "SAT_SYM")])
def test_reg_share_lossy_accumulator_keeps_the_ascending_mac_order(ftype, q, o):
    n_taps = 21 if "ODD" in ftype else 22
    # (TRN into WRAP has no order: the matrix-core class-B kernel; the others keep the exact order)
    want = "mfma_lossy" if (q, o) == ("TRN", "WRAP") else "generic"
    check_reg_share(n_taps, ftype, A.Fmt(14, 4), A.Fmt(12, 2), A.Fmt(20, 8, True, q, o), A.Fmt(10, 5, True, q, o), splits=[7, 300],
                    expect_path=want, seed=5)


@pytest.mark.parametrize("ftype", RS_FTYPES)
@pytest.mark.parametrize("rnd", ["TRN", "RND"])
def test_reg_share_lossy_wrapping_accumulator_on_the_matrix_cores(ftype, rnd):
    """ac_fir_reg_share's cores in class B (ac_fir_reg_share.h:136-260): the anti-symmetric folds subtract the mirrored sample, so the dropped
    bits come from (x[i] - x[N-1-i]) mod 2^s; whole chunks on the ring kernel, ragged calls on the exact-order kernel, odd and even tap counts."""
    for n_taps, fin, fc, fa in ((27 if "ODD" in ftype else 28, A.Fmt(28, 6), A.Fmt(23, 7), A.Fmt(64, 32, True, rnd, "WRAP")),
                                (61 if "ODD" in ftype else 60, A.Fmt(16, 8), A.Fmt(24, 6), A.Fmt(48, 26, True, rnd, "WRAP")),
                                (28 if "ODD" in ftype else 27, A.Fmt(32, 16), A.Fmt(20, 4), A.Fmt(50, 22, True, rnd, "WRAP"))):
        chunk = 4096 if fin.W <= 16 else 2048
        c = rand_raw(np.random.default_rng(n_taps), fc, (n_taps,)) >> max(fc.W - 22, 1)
        check_reg_share(n_taps, ftype, fin, fc, fa, A.Fmt(fa.W, fa.I), n_ch=3, n=3 * chunk + 211, coeffs=c, splits=[2 * chunk, 2 * chunk + 100],
                        expect_path="mfma_lossy", seed=n_taps)


def test_reg_share_rejects_ftypes_without_a_branch():
    for ft in ("ROTATE_SHIFT", "C_BUFF", "TRANSPOSED"):
        with pytest.raises(A.AcdspError):
            A.Fir(8, ft, A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2), kind="reg_share")


@pytest.mark.parametrize("o", ["SAT", "SAT_SYM", "SAT_ZERO"])
def test_saturating_accumulators_that_cannot_saturate_run_the_wrapping_classes(o):
    """AC_SAT* accumulators whose bounds no partial sum of the handle's coefficient set can reach are wrapping accumulators (set_coeffs decides per
    set: acdsp_fir::sat_free).  At the bound exactly -- sum|c| * 2^15 = 2^(W - 1) - 1 is impossible, so one LSB inside and one outside -- with
    full-scale inputs of the signs that reach it; the set that could saturate keeps the exact-order kernels and does saturate."""
    fin, fc = A.Fmt(16, 2), A.Fmt(16, 2)
    n_taps = 32
    c_in = np.array([1023] * 31 + [1024], dtype=np.int64)              # sum|c| = 32737: 32737 * 2^15 <  2^30 - 1
    c_out = np.array([1024] * 32, dtype=np.int64)                       # sum|c| = 32768: 32768 * 2^15 = 2^30  >  2^30 - 1
    fa = A.Fmt(31, 3, True, "TRN", o)                                   # F = 28 = F_in + F_coeff: exact products
    for c, want in ((c_in, "mfma_i8"), (c_out, "generic")):
        for fo in (A.Fmt(16, 3, True, "RND", "SAT"), A.Fmt(31, 3)):
            rng = np.random.default_rng(len(o))
            x = rand_raw(rng, fin, (3, 4096 + 77))
            x[0, :] = -32768                                            # every product at its largest magnitude, same sign
            x[1, ::2] = 32767
            fir = A.Fir(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_channels=3, kind="load")
            fir.set_coeffs(c)
            assert fir.path == want, (fir.path, want)
            y = run_engine(fir, x, [4096])
            yo = OracleFir(n_taps, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=3).run(c, x)
            assert np.array_equal(y, yo), (o, want)
    # the lossy class: two bits dropped per tap, one LSB of slack per tap in the bound
    fa2 = A.Fmt(30, 4, True, "RND", o)
    check_case(63, "SHIFT_REG", fin, fc, fa2, A.Fmt(16, 2, True, "RND", "SAT"), n=2 * 4096, splits=[4096], coeffs=np.round(np.hanning(65)[1:-1] * 1000).astype(np.int64),
               expect_path="mfma_lossy")
    check_case(63, "SHIFT_REG", fin, fc, fa2, A.Fmt(16, 2, True, "RND", "SAT"), n=4096 + 50, splits=[4096], seed=3, expect_path="generic")   # full-range weights: can saturate
    # folds and a set per channel
    check_case(31, "FOLD_ODD", fin, fc, A.Fmt(40, 12, True, "TRN", o), A.Fmt(16, 2, True, "RND", "SAT"), n=4096 + 9, splits=[4096],
               coeffs=np.round(np.hanning(33)[1:-1] * 2000).astype(np.int64), expect_path="mfma_i8")
