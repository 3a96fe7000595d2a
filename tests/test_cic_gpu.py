"""GPU parity tests, CIC decimator / interpolator: HIP engine (C ABI) vs the reference's own exact
vectors and vs the CPU oracle."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from oracle import OracleCic, stimulus
from helpers import ofmt, read_fracs, to_raw

pytestmark = pytest.mark.gpu


def rand_raw(rng, fmt, shape):
    lo = -(1 << (fmt.W - 1)) if fmt.S else 0
    hi = (1 << (fmt.W - 1)) - 1 if fmt.S else (1 << fmt.W) - 1
    return rng.integers(lo, hi + 1, size=shape, dtype=np.int64)


def run_engine(cic, x, splits=None):
    dt = A.torch_dtype_for(cic.fin)
    outs = []
    bounds = [0] + list(splits or []) + [x.shape[1]]
    for a, b in zip(bounds[:-1], bounds[1:]):
        xd = torch.from_numpy(x[:, a:b].copy()).to(dt).cuda()
        if b == a:
            assert cic.out_count(0) == 0
            continue
        outs.append(cic.run(xd).cpu().numpy().astype(np.int64))
    return np.concatenate(outs, axis=1)


def run_oracle(interp, R, M, N, fin, fout, x, splits=None):
    orc = OracleCic(interp, R, M, N, ofmt(fin), ofmt(fout), n_ch=x.shape[0])
    outs = []
    bounds = [0] + list(splits or []) + [x.shape[1]]
    for a, b in zip(bounds[:-1], bounds[1:]):
        outs.append(orc.run(x[:, a:b]))
    return np.concatenate(outs, axis=1)


def test_decimator_reference_vector_exact():
    # tests/rtest_ac_cic_dec_full.cpp: R=7 M=2 N=4, <32,16> -> <48,32>; one extra leading zero (:84-85)
    fin, fout = A.Fmt(32, 16), A.Fmt(48, 32)
    x = np.concatenate([[0], to_raw(read_fracs("ac_cic_dec_full_input.txt"), 16)])[None, :]
    ref = to_raw(read_fracs("ac_cic_dec_full_ref.txt"), 16)
    cic = A.Cic(False, 7, 2, 4, fin, fout)
    assert (cic.int_type.W, cic.int_type.I) == (48, 32)
    y = run_engine(cic, x)[0]
    assert len(y) == 1430 and np.array_equal(y[:len(ref)], ref)
    # same stream in ragged pieces: state carry across run() calls
    cic2 = A.Cic(False, 7, 2, 4, fin, fout)
    y2 = run_engine(cic2, x, splits=[1, 2, 9, 10, 10, 700, 5000])[0]
    assert np.array_equal(y2, y)


def test_interpolator_reference_vector_exact():
    # tests/rtest_ac_cic_intr_full.cpp: R=7 M=2 N=5, 1000 inputs, first N_TB refs discarded (:88,:99)
    fin, fout = A.Fmt(32, 16), A.Fmt(49, 33)
    x = to_raw(read_fracs("ac_cic_intr_full_input.txt"), 16)[:1000][None, :]
    ref = to_raw(read_fracs("ac_cic_intr_full_ref.txt"), 16)[5:]
    cic = A.Cic(True, 7, 2, 5, fin, fout)
    y = run_engine(cic, x)[0]
    assert len(y) == 6990 and np.array_equal(y, ref[:6990])


def test_reference_vectors_many_channels_distinct_delays():
    """The reference's two exact CIC vectors (tests/ac_cic_{dec,intr}_full_{input,ref}.txt) as a bank: channel c carries the testbench stream
    behind its own number of leading zeros, so every channel must reproduce the reference outputs at its own delay -- on the batched kernels
    the bank takes (decimator R = 7: the matrix-core FIR identity; interpolator R = 7: fir_up_kernel's 28-row tiles, whole 512-input steps
    on the matrix cores and the head / tail on the polyphase kernel), one call and ragged calls."""
    n_ch = 48
    # interpolator: R=7 M=2 N=5 (rtest_ac_cic_intr_full.cpp, ac_cic_intr_full_param.h:33-47)
    fin, fout = A.Fmt(32, 16), A.Fmt(49, 33)
    xi = to_raw(read_fracs("ac_cic_intr_full_input.txt"), 16)[:1000]
    ref = to_raw(read_fracs("ac_cic_intr_full_ref.txt"), 16)[5:][:6990]
    n = 1000 + n_ch + 8
    n += (-n) % 16
    x = np.zeros((n_ch, n), dtype=np.int64)
    for c in range(n_ch):
        x[c, c:c + 1000] = xi
    for splits in (None, [600], [33, 34, 1000]):
        cic = A.Cic(True, 7, 2, 5, fin, fout, n_channels=n_ch)
        y = run_engine(cic, x, splits)
        if splits is None:
            assert cic.path == "mfma_gen", cic.path
        for c in range(n_ch):
            assert not y[c, :7 * c].any(), c
            assert np.array_equal(y[c, 7 * c:7 * c + 6990], ref), (c, splits)
    # decimator: R=7 M=2 N=4 (rtest_ac_cic_dec_full.cpp; one extra leading zero :84-85); delays in whole decimation periods keep the phase
    fin, fout = A.Fmt(32, 16), A.Fmt(48, 32)
    xd = np.concatenate([[0], to_raw(read_fracs("ac_cic_dec_full_input.txt"), 16)])
    refd = to_raw(read_fracs("ac_cic_dec_full_ref.txt"), 16)
    n = len(xd) + 7 * n_ch
    n += (-n) % 16
    x = np.zeros((n_ch, n), dtype=np.int64)
    for c in range(n_ch):
        x[c, 7 * c:7 * c + len(xd)] = xd
    for splits in (None, [4096], [7, 700, 5000]):
        cic = A.Cic(False, 7, 2, 4, fin, fout, n_channels=n_ch)
        y = run_engine(cic, x, splits)
        if splits is None:
            assert cic.path == "mfma_gen", cic.path
        for c in range(n_ch):
            assert not y[c, :c].any(), c
            assert np.array_equal(y[c, c:c + len(refd)], refd), (c, splits)


@pytest.mark.parametrize("interp", [False, True])
@pytest.mark.parametrize("R,M,N", [(8, 1, 5), (16, 1, 5), (2, 1, 1), (3, 2, 3), (5, 3, 2), (4, 4, 3), (2, 1, 8), (13, 1, 4)])
def test_parameter_sweep_vs_oracle(interp, R, M, N):
    fin = A.Fmt(32, 16)
    it = A.Cic(interp, R, M, N, fin, fin).int_type
    fout = A.Fmt(it.W, it.I)
    rng = np.random.default_rng(R * 100 + M * 10 + N)
    n_ch, n = 70, 900 if not interp else 300
    x = rand_raw(rng, fin, (n_ch, n))
    splits = [1, 3, 64, 65, 500] if not interp else [1, 2, 100]
    y = run_engine(A.Cic(interp, R, M, N, fin, fout, n_channels=n_ch), x, splits)
    yo = run_oracle(interp, R, M, N, fin, fout, x, splits)
    assert y.shape == yo.shape and np.array_equal(y, yo)


@pytest.mark.parametrize("fin", [A.Fmt(16, 1), A.Fmt(12, 4, False), A.Fmt(36, 21), A.Fmt(24, 8)])
def test_input_containers_and_signedness(fin):
    it = A.Cic(False, 16, 1, 5, fin, fin).int_type
    fout = A.Fmt(it.W, it.I)
    rng = np.random.default_rng(fin.W)
    x = rand_raw(rng, fin, (3, 4000))
    y = run_engine(A.Cic(False, 16, 1, 5, fin, fout, n_channels=3), x, [1000])
    assert np.array_equal(y, run_oracle(False, 16, 1, 5, fin, fout, x, [1000]))


@pytest.mark.parametrize("q,o", [("RND", "SAT"), ("TRN_ZERO", "SAT_SYM"), ("RND_CONV", "WRAP"), ("RND_INF", "SAT_ZERO")])
def test_output_type_conversion(q, o):
    fin, fout = A.Fmt(32, 16), A.Fmt(20, 14, True, q, o)
    rng = np.random.default_rng(3)
    x = rand_raw(rng, fin, (2, 2000))
    y = run_engine(A.Cic(False, 8, 1, 5, fin, fout, n_channels=2), x)
    assert np.array_equal(y, run_oracle(False, 8, 1, 5, fin, fout, x))


def test_many_chunks_time_parallel_equals_serial():
    # long stream: many time chunks per channel, each rebuilt from a zero-state warm-up
    fin = A.Fmt(32, 16)
    fout = A.Fmt(47, 31)
    n_ch, n = 2, 300000
    x = stimulus(0xC1C, n_ch, n, 32)
    y = run_engine(A.Cic(False, 8, 1, 5, fin, fout, n_channels=n_ch), x, [123457])
    assert np.array_equal(y, run_oracle(False, 8, 1, 5, fin, fout, x))


def test_unaligned_input_rows():
    fin, fout = A.Fmt(32, 16), A.Fmt(47, 31)
    x = stimulus(9, 3, 1001, 32)
    big = torch.zeros((3, 1007), dtype=torch.int32, device="cuda")
    big[:, 1:1002] = torch.from_numpy(x).to(torch.int32).cuda()
    cic = A.Cic(False, 8, 1, 5, fin, fout, n_channels=3)
    y = cic.run(big[:, 1:1002]).cpu().numpy().astype(np.int64)
    assert np.array_equal(y, run_oracle(False, 8, 1, 5, fin, fout, x))


def test_rejects_what_the_reference_cannot_compile():
    with pytest.raises(A.AcdspError):
        A.Cic(False, 64, 4, 8, A.Fmt(32, 16), A.Fmt(64, 32))   # (R*M)^N >= 2^31: int power<> overflow
    with pytest.raises(A.AcdspError):
        A.Cic(True, 1, 1, 3, A.Fmt(32, 16), A.Fmt(48, 32))     # R = 1 interpolator never re-arms


def test_config3_sampled_full_rate():
    """BASELINE config 3 shape (N=5 R=8 <32,16>) on 4096 channels; a reduced sample count keeps the test
    inside the test budget -- bench.py runs the 4 Mi-sample size.  Checked on sampled channels."""
    fin, fout = A.Fmt(32, 16), A.Fmt(47, 31)
    n_ch, n = 4096, 1 << 16
    x = torch.empty((n_ch, n), dtype=torch.int32, device="cuda")
    A.fill_stimulus(x, 0xACD5, 32)
    cic = A.Cic(False, 8, 1, 5, fin, fout, n_channels=n_ch)
    y = cic.run(x)
    assert cic.path == "mfma_gen"        # decimator through its FIR identity on the matrix cores
    assert y.shape == (n_ch, n // 8)
    for ch in (0, 63, 64, 2049, 4095):
        yo = run_oracle(False, 8, 1, 5, fin, fout, stimulus(0xACD5, 1, n, 32, ch0=ch))
        assert np.array_equal(y[ch].cpu().numpy().astype(np.int64), yo[0]), ch


def test_recurrence_and_mfma_kernels_agree(monkeypatch):
    """Both decimator kernels (integrator/comb recurrences, FIR identity on MFMA) on the same stream: the
    unaligned view forces the recurrence kernel, the aligned one takes the MFMA kernel."""
    fin, fout = A.Fmt(32, 16), A.Fmt(47, 31)
    x = stimulus(21, 5, 8000, 32)
    big = torch.zeros((5, 8016), dtype=torch.int32, device="cuda")
    big[:, 1:8001] = torch.from_numpy(x).to(torch.int32).cuda()
    c1 = A.Cic(False, 8, 1, 5, fin, fout, n_channels=5)
    y1 = c1.run(big[:, 1:8001]).cpu().numpy()
    assert c1.path == "recurrence"
    c2 = A.Cic(False, 8, 1, 5, fin, fout, n_channels=5)
    xa = torch.zeros((5, 8000), dtype=torch.int32, device="cuda")
    xa.copy_(torch.from_numpy(x).to(torch.int32))
    y2 = c2.run(xa).cpu().numpy()
    assert c2.path == "mfma_gen"
    assert np.array_equal(y1, y2)


@pytest.mark.parametrize("q,o", [("TRN", "WRAP"), ("RND", "WRAP"), ("TRN", "SAT"), ("RND", "SAT"), ("RND_CONV", "SAT_SYM")])
def test_long_runs_take_the_branch_free_kernel(q, o):
    """Runs long enough for whole chunks of complete 256-output steps (fir_gen_fast_kernel) plus a ragged tail
    (general kernel); the last mode pair is outside the fast conversion and must still match."""
    rng = np.random.default_rng(11)
    # BASELINE config 3 shape with a narrowing OUT_TYPE
    fin, fout = A.Fmt(32, 16), A.Fmt(20, 14, True, q, o)
    x = rand_raw(rng, fin, (3, 8 * 5000 + 64))   # bursts of whole 16-sample slots: matrix-core kernels
    cic = A.Cic(False, 8, 1, 5, fin, fout, n_channels=3)
    y = run_engine(cic, x, [8 * 2048])
    assert cic.path == "mfma_gen"
    assert np.array_equal(y, run_oracle(False, 8, 1, 5, fin, fout, x, [8 * 2048]))
    # DDC stage A: R = 16 on ac_fixed<16,1> into the lossless <36,21> (three coefficient digits, six K blocks)
    fin, fout = A.Fmt(16, 1), A.Fmt(36, 21, True, q, o)
    x = rand_raw(rng, fin, (3, 16 * 3000 + 48))
    cic = A.Cic(False, 16, 1, 5, fin, fout, n_channels=3)
    y = run_engine(cic, x)
    assert cic.path == "mfma_gen"
    assert np.array_equal(y, run_oracle(False, 16, 1, 5, fin, fout, x))


@pytest.mark.parametrize("R,M,N,fin,fout", [
    (8, 1, 4, A.Fmt(16, 1), None),                                # int16 samples, INT_TYPE <28,13>: 4-byte containers
    (8, 1, 5, A.Fmt(16, 1), None),                                # ... <31,16>
    (8, 1, 4, A.Fmt(16, 1), A.Fmt(24, 9, True, "RND", "SAT")),    # ... a narrowing conversion
    (4, 1, 3, A.Fmt(32, 16), None),                               # R = 4 on int32: four steps of 4 KB per wave
    (4, 1, 5, A.Fmt(32, 16), None),
    (16, 1, 5, A.Fmt(32, 16), None),                              # R = 16 on int32: three coefficient digits, one 16 KB step per wave
    (16, 1, 5, A.Fmt(32, 16), A.Fmt(40, 20, True, "TRN", "SAT")),
    (7, 2, 4, A.Fmt(32, 16), None),                               # the reference testbench's parameters: odd factor, identity slot map, 7 KB steps
    (7, 2, 4, A.Fmt(32, 16), A.Fmt(48, 32)),
    (7, 2, 4, A.Fmt(32, 16), A.Fmt(40, 24, True, "RND", "SAT")),
    (7, 1, 3, A.Fmt(32, 16), None),
    (7, 2, 4, A.Fmt(30, 10, False), None),                        # unsigned samples: 31-bit signed planes
    (8, 2, 3, A.Fmt(24, 8), None),                                # 24-bit samples in int32 containers: the sign plane rides along as a fourth plane
    (4, 1, 4, A.Fmt(20, 3), A.Fmt(36, 19, True, "RND", "SAT")),
    (10, 1, 5, A.Fmt(32, 16), None),                              # R = 10 (even, does not divide a 1 KB load: identity slot map), two 10 KB steps per wave
    (10, 1, 5, A.Fmt(32, 16), A.Fmt(44, 24, True, "RND", "SAT")),
    (10, 1, 4, A.Fmt(16, 1), None),                               # ... on int16 into 4-byte containers (INT_TYPE <30,15>)
    (10, 2, 3, A.Fmt(16, 1), A.Fmt(40, 25)),                      # ... into 8-byte containers, differential delay 2
    (3, 1, 5, A.Fmt(16, 1), None),                                # small and decimal rates of multistage decimators: 3, 6, 12, 20
    (3, 2, 4, A.Fmt(16, 1), A.Fmt(40, 25)),
    (3, 1, 5, A.Fmt(32, 16), None),
    (6, 1, 5, A.Fmt(16, 1), None),
    (6, 1, 4, A.Fmt(16, 1), A.Fmt(34, 19)),
    (6, 1, 5, A.Fmt(32, 16), A.Fmt(40, 24, True, "RND", "SAT")),
    (12, 1, 4, A.Fmt(16, 1), None),
    (12, 1, 5, A.Fmt(16, 1), None),
    (20, 1, 3, A.Fmt(16, 1), None),
    (20, 1, 5, A.Fmt(16, 1), None),
    (5, 1, 5, A.Fmt(32, 16), None),
    (7, 2, 4, A.Fmt(16, 1), None),                                # the reference testbench's factor on 16-bit samples: INT_TYPE <32,17>, 4-byte containers
    (7, 1, 5, A.Fmt(16, 1), None),                                # ... 16 + 5 log2(7) -> 31 bits, 4-byte containers
    (7, 2, 5, A.Fmt(16, 1), None),                                # ... 36 bits: 8-byte containers
    (5, 1, 6, A.Fmt(16, 1), None),                                # R = 5 on int16: two steps per load group (2.5 KB each), INT_TYPE <30,15>
    (5, 1, 6, A.Fmt(16, 1), A.Fmt(40, 25)),                       # ... into 8-byte containers
    (5, 2, 3, A.Fmt(16, 4), A.Fmt(24, 10, True, "RND", "SAT")),
])
def test_decimator_ring_kernel_shapes_outside_the_baseline(R, M, N, fin, fout):
    """The ring kernel (fir_gen_ring_kernel) serves more CIC decimator shapes than the BASELINE ones: whole chunks on it, the ragged
    tail and the continuation calls (history in front of the first window) on the general kernel; every output against the oracle."""
    rng = np.random.default_rng(7 * R + N)
    probe = A.Cic(False, R, M, N, fin, fin)
    it = probe.int_type
    fo = fout if fout is not None else A.Fmt(it.W, it.I)
    n = R * (256 * 4 * 3 + 16 * 5) + 16                           # three chunks of four steps + a ragged tail, whole 16-sample slots
    x = rand_raw(rng, fin, (3, n))
    x[1, :R * 300] = (1 << (fin.W - 1)) - 1
    x[2, :R * 300] = -(1 << (fin.W - 1))
    for splits in (None, [R * 256 * 5 + 16 * R]):
        cic = A.Cic(False, R, M, N, fin, fo, n_channels=3)
        y = run_engine(cic, x, splits)
        assert cic.path == "mfma_gen"
        yo = run_oracle(False, R, M, N, fin, fo, x, splits)
        assert y.shape == yo.shape
        bad = np.argwhere(y != yo)
        assert bad.size == 0, "%d mismatches, first at %s" % (len(bad), bad[0])


# ---- interpolator on the matrix cores (fir_up.hip): whole steps of 512 inputs; head and tail on the polyphase VALU kernel ----

@pytest.mark.parametrize("R,M,N,fin,fout", [
    (8, 1, 5, A.Fmt(32, 16), None),                               # bench shape: INT_TYPE <44,28> out (bit-field wrap epilogue)
    (8, 1, 5, A.Fmt(32, 16), A.Fmt(40, 24)),                      # narrower signed OUT_TYPE, still wider than 32 bits
    (8, 1, 5, A.Fmt(32, 16), A.Fmt(36, 20, False)),               # unsigned OUT_TYPE narrower than INT_TYPE (mask)
    (8, 1, 5, A.Fmt(32, 16), A.Fmt(50, 34, False)),               # unsigned and wider than INT_TYPE: generic conversion
    (8, 1, 5, A.Fmt(32, 16), A.Fmt(24, 8, True, "RND", "SAT")),   # int32 samples into 4-byte containers: not compiled in, VALU kernel
    (16, 1, 4, A.Fmt(16, 1), None),                               # int16 samples, R = 16
    (8, 1, 5, A.Fmt(16, 1), A.Fmt(16, 1, True, "TRN", "WRAP")),   # int16 samples into 2-byte containers (generic conversion)
    (16, 1, 4, A.Fmt(16, 1), A.Fmt(16, 1, True, "RND", "SAT")),   # N - 1 = 3 skipped 2-byte outputs: unaligned runs, VALU kernel
    (4, 2, 5, A.Fmt(16, 4), None),                                # R = 4, differential delay 2
    (8, 2, 4, A.Fmt(30, 10, False), None),                        # unsigned input: 31-bit signed planes
    (8, 1, 5, A.Fmt(16, 1), None),                                # int16 samples, INT_TYPE <28,13>: 4-byte containers, 32-bit wrap epilogue
    (4, 1, 5, A.Fmt(16, 1), None),                                # ... R = 4: four steps per wave
    (8, 1, 5, A.Fmt(16, 1), A.Fmt(20, 5, False)),                 # ... unsigned OUT_TYPE narrower than INT_TYPE (mask)
    (8, 1, 5, A.Fmt(16, 1), A.Fmt(32, 17)),                       # ... signed OUT_TYPE wider than INT_TYPE, full container
    (8, 1, 5, A.Fmt(16, 1), A.Fmt(24, 9, True, "RND", "SAT")),    # ... a real conversion into 4-byte containers (generic epilogue)
    (16, 1, 5, A.Fmt(16, 1), None),                               # boxcar(16)^5 taps pass 2^15: three digit planes; INT_TYPE exactly 32 bits
    (16, 1, 5, A.Fmt(32, 16), None),                              # three digit planes on int32 samples, INT_TYPE <48,32>
    (16, 1, 5, A.Fmt(32, 16), A.Fmt(40, 24, False)),              # ... unsigned narrower OUT_TYPE
    (7, 2, 5, A.Fmt(32, 16), None),                               # the reference testbench's parameters: 28 live rows per tile
    (7, 2, 5, A.Fmt(32, 16), A.Fmt(49, 33)),
    (7, 2, 5, A.Fmt(16, 1), None),                                # ... on int16 samples: INT_TYPE <36,21>
    (7, 1, 4, A.Fmt(16, 1), None),                                # INT_TYPE <26,11>: 4-byte containers
    (3, 1, 4, A.Fmt(16, 2), None),                                # R = 3: 8 samples x 3 phases = 24 rows
    (5, 2, 3, A.Fmt(32, 16), None),                               # R = 5: 20 rows
    (6, 1, 5, A.Fmt(16, 1), None),                                # R = 6: 24 rows
    (6, 1, 5, A.Fmt(32, 16), A.Fmt(40, 24, False)),
])
def test_interpolator_matrix_core_path(R, M, N, fin, fout):
    rng = np.random.default_rng(R * 100 + N)
    probe = A.Cic(True, R, M, N, fin, fin)
    it = probe.int_type
    fo = fout if fout is not None else A.Fmt(it.W, it.I)
    n = 16 * 34 + 512 * 2 + 48                                   # rows of whole 16-byte pieces
    x = rand_raw(rng, fin, (3, n))
    x[1, :64] = (1 << (fin.W - 1)) - 1 if fin.S else (1 << fin.W) - 1
    x[2, :64] = -(1 << (fin.W - 1)) if fin.S else 0
    dt_i, dt_o = A.torch_dtype_for(fin), A.torch_dtype_for(fo)
    want_mfma = dt_o == torch.int64 or (dt_i == torch.int16 and dt_o == torch.int32) or (dt_o == torch.int16 and dt_i == torch.int16 and N == 5)
    for splits in (None, [600], [16, 1200], [5, 1205]):
        cic = A.Cic(True, R, M, N, fin, fo, n_channels=3)
        y = run_engine(cic, x, splits)
        yo = run_oracle(True, R, M, N, fin, fo, x, splits)
        assert y.shape == yo.shape
        bad = np.argwhere(y != yo)
        assert bad.size == 0, "%d mismatches, first at %s" % (len(bad), bad[0])
        if splits is None:
            assert cic.path == ("mfma_gen" if want_mfma else "fir_identity"), cic.path


def test_interpolator_both_kernels_agree(monkeypatch):
    fin = A.Fmt(32, 16)
    fo = A.Fmt(44, 28)
    x = stimulus(5, 4, 16 * 34 + 512 * 5, 32)
    xa = torch.from_numpy(x).to(torch.int32).cuda()
    c1 = A.Cic(True, 8, 1, 5, fin, fo, n_channels=4)
    y1 = c1.run(xa).cpu().numpy()
    assert c1.path == "mfma_gen"
    big = torch.zeros((4, x.shape[1] + 16), dtype=torch.int32, device="cuda")
    big[:, 1:1 + x.shape[1]] = xa
    c2 = A.Cic(True, 8, 1, 5, fin, fo, n_channels=4)
    y2 = c2.run(big[:, 1:1 + x.shape[1]]).cpu().numpy()      # unaligned rows: polyphase VALU kernel
    assert c2.path == "fir_identity"
    assert np.array_equal(y1, y2)
