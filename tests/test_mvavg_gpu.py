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


def check(taps, mode, fin, fc, fa, fo, n_sample, n_frames, n_obj=3, seed=0, force_generic=False, max_sample=None, coeffs=None, path=None):
    rng = np.random.default_rng(seed)
    x = rand_raw(rng, fin, (n_obj, n_sample * n_frames))
    c = rand_raw(rng, fc, (taps,)) if coeffs is None else np.asarray(coeffs, dtype=np.int64)
    eng = A.MvAvg(max_sample or max(n_sample, 1), taps, mode, fin, fc, fa, fo, n_objects=n_obj, force_generic=force_generic)
    eng.set_coeffs(c)
    y = eng.run(torch.from_numpy(x).to(A.torch_dtype_for(fin)).cuda(), n_sample).cpu().numpy().astype(np.int64)
    yo = OracleMvAvg(taps, mode, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_obj=n_obj).run(c, x, n_sample)
    assert y.shape == yo.shape, (y.shape, yo.shape)
    bad = np.argwhere(y != yo)
    assert bad.size == 0, "%d mismatches, first at %s" % (len(bad), bad[0])
    if path is not None:
        # "stream": the streaming kernel in either form (v_dot2 / matrix cores -- the split is by window length, tested in
        # test_stream_kernel_classes); "stream_dot2" / "stream_mfma": that form
        # "int64_sums": the general order-free kernels -- since round 6 the sliding-window kernel takes 16-bit samples too where its class conditions hold
        allowed = {"stream": ("stream", "stream_mfma"), "stream_dot2": ("stream",), "int64_sums": ("int64_sums", "stream32")}.get(path, (path,))
        assert eng.path in allowed, (eng.path, path)


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


def small_coeffs(rng, taps, total=32767):
    """TAPS signed 16-bit weights whose absolute values sum to <= total (the streaming kernel's int32 bound)."""
    c = rng.integers(-total // taps, total // taps + 1, size=taps, dtype=np.int64)
    c[taps // 2] += (total - np.abs(c).sum()) * (1 if rng.integers(0, 2) else -1)
    assert np.abs(c).sum() <= total
    return c


STREAM_ACC = [
    (A.Fmt(40, 18), "linear, d = sh"),                  # F_acc = F_in + F_coeff: exact products
    (A.Fmt(48, 20), "linear, d > sh"),                  # products shifted left
    (A.Fmt(32, 16), "per tap, TRN"),                    # F_acc < F_in + F_coeff: every product loses bits
    (A.Fmt(32, 16, True, "RND"), "per tap, RND"),
    (A.Fmt(30, 14, True, "RND"), "per tap, 64-bit epilogue"),   # W_acc < 31: the accumulator may wrap
]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("taps", [1, 3, 9, 11, 17, 25, 33, 35, 49, 51, 65])
def test_stream_kernel_classes(mode, taps):
    """16-bit samples / weights with sum |c| < 2^15, frames aligned to 16 bytes: the streaming kernel (path 'stream')."""
    rng = np.random.default_rng(100 + taps)
    fin, fc = A.Fmt(16, 8), A.Fmt(16, 2)
    for k, (fa, _) in enumerate(STREAM_ACC):
        for fo in (A.Fmt(16, 8, True, "RND", "SAT"), A.Fmt(24, 10, True, "TRN", "WRAP"), A.Fmt(40, 18), A.Fmt(12, 6, False, "RND", "SAT")):
            # windows of 11 taps and more (9 with AC_WIN) in the linear class: the same kernel with its sums on the matrix cores (round 6)
            # (AC_WIN at 9 taps and 1016 outputs per frame: the output-row walk of the v_dot2 form)
            want = "stream_mfma" if (taps >= 11 and k < 2) else "stream_dot2"
            check(taps, mode, fin, fc, fa, fo, 1024, 3, n_obj=2, seed=200 + k, coeffs=small_coeffs(rng, taps), path=want)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("taps", [11, 13, 15, 17, 19, 21, 23, 27, 31, 33, 35, 41, 47, 49, 57, 63, 65])
def test_window_sums_on_the_matrix_cores(mode, taps):
    """Round 6: 11 taps and more, linear class.  Every odd window offset into the image (TAPS / 2 mod 8), one and two K blocks, frame lengths around
    the 512-output tile, full-scale samples against extreme weights, both epilogues, all container widths of the output."""
    rng = np.random.default_rng(300 + taps)
    fin, fc = A.Fmt(16, 8), A.Fmt(16, 2)
    for fa, fo in ((A.Fmt(40, 18), A.Fmt(16, 8, True, "RND", "SAT")), (A.Fmt(48, 20), A.Fmt(24, 10, True, "TRN", "WRAP")), (A.Fmt(40, 18), A.Fmt(40, 18)),
                   (A.Fmt(30, 8), A.Fmt(12, 6, False, "RND", "SAT"))):       # last: W_acc = 30 may wrap -- the 64-bit epilogue
        for n in (72, 504, 512, 520, 1024, 1032, 4096 + 8):
            if n >= taps:
                check(taps, mode, fin, fc, fa, fo, n, 3, n_obj=2, seed=n + fa.W, coeffs=small_coeffs(rng, taps), path="stream_mfma", max_sample=8192)
    fa, fo = A.Fmt(40, 18), A.Fmt(40, 18)
    # extreme weights and samples: one weight of 32639 (the largest whose balanced high digit is a signed byte), alternating +-, a negative giant
    for k, c in enumerate(([0] * (taps // 2) + [32639] + [0] * (taps // 2), [(-1) ** j * (32767 // taps) for j in range(taps)],
                           [-32767] + [0] * (taps - 1), [0] * (taps - 1) + [-32640], [127] * taps, [-128] * taps, [128] * taps, [-129] * taps)):
        rs = np.random.default_rng(k)
        n_sample, n_frames = 1024, 4
        x = rs.choice(np.array([-32768, 32767, -1, 0, 255, 256, -256, 127, 128, -129], dtype=np.int64), size=(2, n_sample * n_frames))
        eng = A.MvAvg(n_sample, taps, mode, fin, fc, fa, fo, n_objects=2)
        eng.set_coeffs(np.asarray(c, dtype=np.int64))
        y = eng.run(torch.from_numpy(x).to(torch.int16).cuda(), n_sample).cpu().numpy().astype(np.int64)
        assert eng.path == "stream_mfma", (k, eng.path)
        yo = OracleMvAvg(taps, mode, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_obj=2).run(np.asarray(c, dtype=np.int64), x, n_sample)
        assert np.array_equal(y, yo), (k, np.argwhere(y != yo)[:4])
    # a weight above 32639 keeps the v_dot2 form
    check(taps, mode, fin, fc, fa, fo, 1024, 2, coeffs=[0] * (taps - 1) + [32700], seed=5, path="stream_dot2")


@pytest.mark.parametrize("taps", [3, 5, 7, 11, 13, 21, 35, 63])
def test_output_frames_off_a_16_byte_boundary_leave_as_aligned_runs(taps):
    """AC_WIN with TAPS - 1 no multiple of 8: N - TAPS + 1 outputs per frame, so every frame of the output row starts at another offset inside a
    16-byte granule (round 6: the tile goes through the wave's image and leaves in aligned pieces; first / last piece by elements).  2-, 4- and
    8-byte output containers, frames around the tile size, an output row that itself starts off the boundary."""
    rng = np.random.default_rng(taps)
    fin, fc, fa = A.Fmt(16, 8), A.Fmt(16, 2), A.Fmt(40, 18)
    for fo in (A.Fmt(16, 8, True, "RND", "SAT"), A.Fmt(32, 12), A.Fmt(40, 18)):
        for n in (72, 512, 520, 1024, 1032, 2048 + 8):
            if n >= taps:
                check(taps, "WIN", fin, fc, fa, fo, n, 7, n_obj=3, seed=n, coeffs=small_coeffs(rng, taps), path="stream", max_sample=4096)
    # the output row starts 2 / 6 / 14 bytes into a granule (a view into a larger buffer), rows of an odd stride
    fo = A.Fmt(16, 8, True, "RND", "SAT")
    c = small_coeffs(rng, taps)
    for mode, shift in (("WIN", 1), ("MIRROR", 3), ("CLIP", 7)):
        n, nf, n_obj = 1024, 5, 3
        x = rand_raw(rng, fin, (n_obj, n * nf))
        eng = A.MvAvg(n, taps, mode, fin, fc, fa, fo, n_objects=n_obj)
        eng.set_coeffs(c)
        opf = eng.out_per_frame(n)
        buf = torch.full((n_obj, opf * nf + 21), -7, dtype=torch.int16, device="cuda")
        out = buf[:, shift:shift + opf * nf]
        y = eng.run(torch.from_numpy(x).to(torch.int16).cuda(), n, out=out)
        assert y.data_ptr() == out.data_ptr()
        yo = OracleMvAvg(taps, mode, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_obj=n_obj).run(c, x, n)
        assert np.array_equal(y.cpu().numpy().astype(np.int64), yo), (mode, shift)
        full = buf.cpu().numpy()
        assert (full[:, :shift] == -7).all() and (full[:, shift + opf * nf:] == -7).all(), "stores outside the output row"
        assert eng.path in ("stream", "stream_mfma")


@pytest.mark.parametrize("mode", ["MIRROR", "CLIP"])
@pytest.mark.parametrize("taps", [1, 3, 5, 9])
def test_frames_that_are_no_multiple_of_the_tile_walk_the_row(mode, taps):
    """Round 6: with AC_CLIP / AC_MIRROR and frames of at least 512 + 2 hb samples that are no multiple of 512, the tiles walk the row in aligned
    512-output steps and the frame edges fall anywhere inside them (both halos of an edge in a gap of the wave's image).  Frame lengths that put
    the edge at every kind of place -- just behind a tile start, just in front of a tile end, inside the left / right margin of the image --
    single-frame rows, rows whose length is no multiple of 512, many objects."""
    rng = np.random.default_rng(taps)
    fin, fc, fa = A.Fmt(16, 8), A.Fmt(16, 2), A.Fmt(40, 18)
    for fo in (A.Fmt(16, 8, True, "RND", "SAT"), A.Fmt(32, 12)):
        for n, nf in ((1000, 9), (528, 7), (536, 33), (600, 5), (1016, 4), (1032, 4), (1040, 3), (2056, 3), (4104, 2), (1000, 1), (520, 6), (7 * 512 + 8, 5), (768, 10)):
            check(taps, mode, fin, fc, fa, fo, n, nf, n_obj=3, seed=n + nf, coeffs=small_coeffs(rng, taps), path="stream", max_sample=8192)
    # edges at every residue of 8 inside a tile: frames of 512 + 8 j samples
    for j in (3, 9, 17, 31, 47, 63):
        check(taps, mode, fin, fc, fa, A.Fmt(16, 8, True, "RND", "SAT"), 512 + 8 * j, 13, n_obj=2, seed=j, coeffs=small_coeffs(rng, taps), path="stream", max_sample=8192)
    check(taps, mode, fin, fc, fa, A.Fmt(16, 8, True, "RND", "SAT"), 1000, 70, n_obj=40, seed=99, coeffs=small_coeffs(rng, taps), path="stream", max_sample=8192)


@pytest.mark.parametrize("taps", [9, 17, 25, 33])
def test_ac_win_frames_walk_the_output_row(taps):
    """Round 6: AC_WIN with TAPS - 1 a multiple of 8 and at least 512 outputs per frame -- the tiles walk the OUTPUT row in aligned 512-output steps,
    the windows behind a frame edge start TAPS - 1 samples further on in the contiguous input image (ROW 2; by default for 9 taps, longer windows
    with ACDSP_MVAVG_ROWW_MAX -- the matrix-core form otherwise: either way the oracle's outputs).  Edges at every kind of place inside a tile."""
    rng = np.random.default_rng(taps)
    fin, fc, fa = A.Fmt(16, 8), A.Fmt(16, 2), A.Fmt(40, 18)
    for fo in (A.Fmt(16, 8, True, "RND", "SAT"), A.Fmt(32, 12), A.Fmt(40, 18)):
        for n, nf in ((1024, 9), (1000, 7), (520 + taps - 1, 6), (528 + taps - 1, 33), (600, 5), (1536, 4), (2056, 3), (4096, 3), (1024, 1), (768, 10)):
            check(taps, "WIN", fin, fc, fa, fo, n, nf, n_obj=3, seed=n + nf, coeffs=small_coeffs(rng, taps), path="stream", max_sample=8192)
    for j in (1, 9, 17, 31, 47, 63):
        check(taps, "WIN", fin, fc, fa, A.Fmt(16, 8, True, "RND", "SAT"), 512 + 8 * j + taps - 1, 13, n_obj=2, seed=j, coeffs=small_coeffs(rng, taps), path="stream", max_sample=8192)
    check(taps, "WIN", fin, fc, fa, A.Fmt(16, 8, True, "RND", "SAT"), 1024, 70, n_obj=40, seed=98, coeffs=small_coeffs(rng, taps), path="stream", max_sample=8192)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("taps", [3, 9, 17, 33, 65])
def test_signed_16_bit_saturating_outputs_take_the_packed_epilogue(mode, taps):
    """Round 6: OUT_TYPEs of 16 signed bits with AC_SAT leave the 32-bit epilogue through v_cvt_pk_i16_i32 (shift, then a saturating pack of two outputs).
    Sums that exceed both ends of the output range by far and by one LSB, AC_TRN and AC_RND, every shift the class admits, v_dot2 and matrix-core forms."""
    fin, fc = A.Fmt(16, 8), A.Fmt(16, 2)
    rng = np.random.default_rng(taps)
    c_big = np.full(taps, 32767 // taps, dtype=np.int64)                 # gain ~2: full-scale inputs saturate the output both ways
    c_edge = np.zeros(taps, dtype=np.int64)
    c_edge[taps // 2] = 16384                                              # gain exactly 1: outputs AT the ends of the range
    for fa, fo in ((A.Fmt(40, 18), A.Fmt(16, 8, True, "RND", "SAT")), (A.Fmt(40, 18), A.Fmt(16, 8, True, "TRN", "SAT")), (A.Fmt(40, 18), A.Fmt(16, 2, True, "RND", "SAT")),
                   (A.Fmt(48, 20), A.Fmt(16, 12, True, "RND", "SAT")), (A.Fmt(40, 18), A.Fmt(16, 9, True, "TRN", "SAT"))):
        for c in (c_big, c_edge, -c_big, small_coeffs(rng, taps)):
            n_sample, n_frames = 1024, 3
            x = rng.choice(np.array([-32768, 32767, -32767, 32766, 0, 1, -1, 12345, -23456], dtype=np.int64), size=(2, n_sample * n_frames))
            x[0, :700] = 32767
            x[0, 700:1500] = -32768
            eng = A.MvAvg(n_sample, taps, mode, fin, fc, fa, fo, n_objects=2)
            eng.set_coeffs(c)
            y = eng.run(torch.from_numpy(x).to(torch.int16).cuda(), n_sample).cpu().numpy().astype(np.int64)
            yo = OracleMvAvg(taps, mode, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_obj=2).run(c, x, n_sample)
            assert np.array_equal(y, yo), (fa, fo, np.argwhere(y != yo)[:4])
            assert eng.path in ("stream", "stream_mfma")


def test_matrix_core_window_sums_many_frames_and_objects():
    """Runs of tiles that cross frames and objects, a partial last run, unsigned 15-bit samples."""
    rng = np.random.default_rng(77)
    fc, fa, fo = A.Fmt(16, 2), A.Fmt(40, 18), A.Fmt(16, 8, True, "RND", "SAT")
    check(33, "MIRROR", A.Fmt(16, 8), fc, fa, fo, 1024, 37, n_obj=7, seed=1, coeffs=small_coeffs(rng, 33), path="stream_mfma")
    check(65, "CLIP", A.Fmt(16, 8), fc, fa, fo, 520, 61, n_obj=3, seed=2, coeffs=small_coeffs(rng, 65), path="stream_mfma")
    check(25, "WIN", A.Fmt(15, 8, False), fc, fa, A.Fmt(15, 7, False, "RND", "SAT"), 2048, 9, n_obj=5, seed=3, coeffs=small_coeffs(rng, 25), path="stream_mfma")
    check(33, "MIRROR", A.Fmt(12, 4), fc, A.Fmt(32, 10), A.Fmt(12, 4, True, "RND", "SAT"), 1024, 9, n_obj=5, seed=4, coeffs=small_coeffs(rng, 33), path="stream_mfma")


@pytest.mark.parametrize("mode", MODES)
def test_stream_kernel_shapes(mode):
    """Frame lengths around the 512-output tile, single-tile frames, many frames per wave, unaligned output frames."""
    rng = np.random.default_rng(7)
    fin, fc, fa, fo = A.Fmt(16, 8), A.Fmt(16, 2), A.Fmt(40, 18), A.Fmt(16, 8, True, "RND", "SAT")
    for n in (8, 16, 40, 504, 512, 520, 1016, 1024, 1032, 2048, 4096 + 8):
        for taps in (5, 9):
            if n >= taps:
                check(taps, mode, fin, fc, fa, fo, n, 5, n_obj=3, seed=n, coeffs=small_coeffs(rng, taps), path="stream", max_sample=8192)
    check(9, mode, fin, fc, fa, fo, 64, 700, n_obj=5, seed=1, coeffs=small_coeffs(rng, 9), path="stream")       # runs of tiles cross frames and objects
    check(33, mode, fin, fc, fa, A.Fmt(40, 18), 256, 9, n_obj=2, seed=2, coeffs=small_coeffs(rng, 33), path="stream")   # packed short frames keep v_dot2
    # extreme samples and weights: the int32 bound is tight
    c = np.zeros(9, dtype=np.int64)
    c[4] = 32767
    check(9, mode, A.Fmt(16, 8), fc, fa, fo, 1024, 2, coeffs=c, seed=3, path="stream")
    c[:] = [-4095, 4095, -4095, 4095, -4095, 4095, -4095, 4095, -7]
    check(9, mode, A.Fmt(16, 8), fc, fa, fo, 1024, 2, coeffs=c, seed=4, path="stream")


@pytest.mark.parametrize("mode", MODES)
def test_stream_kernel_narrow_sample_types_keep_the_32_bit_epilogue(mode):
    """Samples of fewer than 16 bits bound the sums lower: an accumulator sized for THEM (W_acc < 31) cannot wrap either and keeps the 32-bit
    conversion; one bit less and it may wrap (64-bit epilogue, same results)."""
    rng = np.random.default_rng(12)
    for fin, fa in ((A.Fmt(12, 4), A.Fmt(30, 10)), (A.Fmt(12, 4), A.Fmt(27, 7)), (A.Fmt(12, 4), A.Fmt(26, 6)), (A.Fmt(10, 2, False), A.Fmt(26, 4)),
                    (A.Fmt(8, 1), A.Fmt(23, 2))):
        for fo in (A.Fmt(12, 4, True, "RND", "SAT"), A.Fmt(16, 4, True, "TRN", "WRAP")):
            check(9, mode, fin, A.Fmt(16, 2), fa, fo, 1024, 3, n_obj=3, seed=fa.W, coeffs=small_coeffs(rng, 9), path="stream")


def test_stream_kernel_fallbacks():
    """Shapes / types outside the streaming class run on the general kernels with the same results."""
    rng = np.random.default_rng(9)
    fin, fc, fa, fo = A.Fmt(16, 8), A.Fmt(16, 2), A.Fmt(40, 18), A.Fmt(16, 8, True, "RND", "SAT")
    c = small_coeffs(rng, 9)
    check(9, "MIRROR", fin, fc, fa, fo, 1020, 3, coeffs=c, seed=1, path="int64_sums")                 # frames not 16-byte aligned
    check(9, "MIRROR", fin, fc, fa, fo, 1024, 3, coeffs=[12000, -9000] * 4 + [30000], seed=2, path="int64_sums")             # sum |c| >= 2^15
    check(9, "MIRROR", fin, fc, A.Fmt(40, 18, True, "TRN", "SAT"), fo, 1024, 3, coeffs=c, seed=3, path="stream")           # saturating, but sum|c| 2^15 is far inside 39 bits: a wrapping one
    check(9, "MIRROR", fin, fc, A.Fmt(24, 9, True, "TRN", "SAT_SYM"), fo, 1024, 3, coeffs=c, seed=13, path="stream")        # ... 32767 * 2^15 * 2^7 / 2^14 + 10 = 8388362 <= 2^23 - 1: just inside
    check(9, "MIRROR", fin, fc, A.Fmt(23, 8, True, "TRN", "SAT"), fo, 1024, 3, coeffs=[3600] * 9, seed=14, path="exact_order")   # 32400 * 2^15 > 2^22 * 2^7: can saturate
    check(9, "MIRROR", fin, fc, A.Fmt(20, 4), fo, 1024, 3, coeffs=c, seed=4, path="int64_sums")       # the cast to ACC wraps
    check(9, "WIN", A.Fmt(16, 8, False), fc, fa, fo, 1024, 3, coeffs=c, seed=5, path="int64_sums")    # unsigned 16-bit samples
    check(9, "WIN", A.Fmt(15, 8, False), fc, fa, fo, 1024, 3, coeffs=c, seed=6, path="stream")        # unsigned 15-bit: fits int16
    check(5, "WIN", fin, fc, fa, fo, 1024, 3, coeffs=small_coeffs(rng, 5), seed=7, path="stream")      # 1020 outputs per frame: element stores


@pytest.mark.parametrize("mode", MODES)
def test_32_bit_samples_on_the_sliding_window_kernel(mode):
    """int32 containers whose cast to ACC_TYPE is exact, coefficients inside int32, wrapping AC_TRN / AC_RND accumulator (mv_avg_w32_kernel):
    whole tiles of 1024 outputs, ragged frames, frames shorter than a tile, windows longer than one 16-byte group, each OUT container."""
    f32 = A.Fmt(32, 16)
    check(9, mode, f32, A.Fmt(16, 2), A.Fmt(56, 30), A.Fmt(32, 16, True, "RND", "SAT"), 2048, 3, seed=1, path="stream32")     # linear: d = sh
    check(9, mode, f32, A.Fmt(16, 2), A.Fmt(62, 22), A.Fmt(62, 22), 1024, 2, seed=2, path="stream32")                          # d = 24 > sh = 14: products shifted left by ten
    check(9, mode, f32, A.Fmt(16, 2), A.Fmt(48, 22, True, "RND"), A.Fmt(16, 4, True, "RND", "SAT"), 1500, 3, seed=3, path="stream32")   # per-tap rounding, 2-byte OUT
    check(33, mode, A.Fmt(24, 8), A.Fmt(24, 2), A.Fmt(40, 16), A.Fmt(40, 16), 1030, 2, seed=4, path="stream32")                # wrapping sums, ragged frame
    check(65, mode, A.Fmt(20, 4, False), A.Fmt(18, 1), A.Fmt(44, 18), A.Fmt(30, 10, False), 300, 5, seed=5, path="stream32")   # unsigned samples, short frames
    check(3, mode, f32, A.Fmt(16, 2), A.Fmt(56, 30), A.Fmt(64, 32), 4096, 2, seed=6, path="stream32")
    check(1, mode, f32, A.Fmt(16, 2), A.Fmt(56, 30), A.Fmt(32, 16), 1024, 2, seed=7, path="stream32")
    check(9, mode, f32, A.Fmt(16, 2), A.Fmt(56, 30, True, "TRN", "SAT"), A.Fmt(32, 16), 1024, 2, seed=8, path="stream32")      # saturating accumulator no sum can reach: a wrapping one
    check(9, mode, f32, A.Fmt(16, 2), A.Fmt(40, 14, True, "TRN", "SAT"), A.Fmt(32, 14), 1024, 2, seed=18, path="exact_order")  # ... whose cast of a sample already saturates: per-tap order
    check(9, mode, f32, A.Fmt(32, 16), A.Fmt(64, 32), A.Fmt(64, 32), 1024, 2, seed=9)                                          # 64-bit products: not this class


def test_usage_example_types():
    """The header's usage example (ac_mv_avg.h:47-51): IN <32,16>, OUT <64,32>, ACC <16,2> (the cast of a sample drops bits and wraps), COEFF
    <32,16> -- the order-free int64 kernel."""
    check(9, "MIRROR", A.Fmt(32, 16), A.Fmt(32, 16), A.Fmt(16, 2), A.Fmt(64, 32), 1024, 3, seed=11, path="int64_sums")


@pytest.mark.parametrize("mode", ["MIRROR", "CLIP"])
@pytest.mark.parametrize("n_sample", [64, 128, 256])
def test_short_frames_share_a_tile(mode, n_sample):
    """Frames of 64 / 128 / 256 samples are packed 8 / 4 / 2 to a 512-output tile of the streaming kernel (own image slot and patched halos per
    frame); frame counts that are no multiple of that, and AC_WIN, keep one frame per tile.  Every window length class, every OUT container."""
    rng = np.random.default_rng(n_sample)
    fin, fc, fa = A.Fmt(16, 8), A.Fmt(16, 2), A.Fmt(40, 18)
    per = 512 // n_sample
    for taps, fo, nf in ((9, A.Fmt(16, 8, True, "RND", "SAT"), 8 * per), (33, A.Fmt(32, 12), 3 * per), (63 if n_sample > 64 else 31, A.Fmt(40, 18), 2 * per),
                         (1, A.Fmt(16, 8, True, "RND", "SAT"), per), (17, A.Fmt(16, 8, True, "TRN", "WRAP"), 5 * per + 1), (5, A.Fmt(24, 10), 40 * per)):   # 5 per + 1 frames: not packed, one frame per tile
        check(taps, mode, fin, fc, fa, fo, n_sample, nf, n_obj=3, seed=taps + nf, coeffs=small_coeffs(rng, taps), path="stream")
    check(9, "WIN", fin, fc, fa, A.Fmt(16, 8, True, "RND", "SAT"), n_sample, 4 * per, coeffs=small_coeffs(rng, 9), seed=3)
    check(9, mode, A.Fmt(12, 4), fc, A.Fmt(24, 10), A.Fmt(12, 4, True, "RND", "SAT"), n_sample, 6 * per, coeffs=small_coeffs(rng, 9), seed=4, path="stream")   # per-tap class
