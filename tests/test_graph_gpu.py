"""GPU tests: run() calls captured into a HIP graph (torch.cuda.CUDAGraph) and replayed.

Small chunks are launch-bound (a 255-tap step over 256 ch x 4096 samples is ~10 us of GPU work behind two launches), so a
streaming deployment captures a fixed schedule of run() calls once and replays it.  run() performs no
allocation, synchronisation or host copy in steady state, so it is capturable; the handle's host-side bookkeeping must be
the same after the captured sequence as before it: calls of at least `hl` samples (the handle's history length: N_TAPS - 1
rounded up to a multiple of 32, the window of the class for CIC / DDC) update the history in place, so any number of them
qualifies; every captured decimator call must consume a multiple of the rate change (acdsp_cic_run / acdsp_ddc_run refuse other
lengths while capturing); shorter calls flip a double buffer and have to come in pairs (engine.hip: hist_next_index).  Replays
continue the stream: the state lives in device memory and is carried from replay to replay exactly as from call to call
(ac_fir_load_coeffs.h:180-188, ac_cic_full_core.h:71-74)."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A

pytestmark = pytest.mark.gpu


def _capture_and_check(make, x, out_shape, out_dtype, n_replays=3):
    """make() -> fresh engine; x [nk][nch][cs] device inputs.  Eager reference: one engine fed the chunks in order, n_replays
    times over; graph: the nk calls captured once, replayed n_replays times."""
    nk = x.shape[0]
    ref_eng = make()
    refs = [torch.stack([ref_eng.run(x[k]).clone() for k in range(nk)]) for _ in range(n_replays)]
    torch.cuda.synchronize()
    eng = make()
    y = torch.zeros((nk,) + out_shape, dtype=out_dtype, device="cuda")
    eng.run(x[0], out=y[0])            # one-time uploads (fragments per phase) happen outside the capture
    eng.run(x[1], out=y[1])
    eng.reset()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.Stream()):
        for k in range(nk):
            eng.run(x[k], out=y[k])
    eng.reset()                        # whatever ran during capture set-up: start the stream from the constructed state
    torch.cuda.synchronize()
    for r in range(n_replays):
        y.zero_()
        g.replay()
        torch.cuda.synchronize()
        got = y[:, :, :refs[r].shape[2]]
        assert torch.equal(got, refs[r]), "replay %d differs from the eager stream" % r


@pytest.mark.parametrize("n_taps", [63, 255, 1023])
def test_fir_calls_replayed_from_a_graph_continue_the_stream(n_taps):
    nch, cs, nk = 48, 4096, 3          # an odd number of calls: long calls update the history in place
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(42, 14), A.Fmt(16, 2, True, "RND", "SAT")
    rng = np.random.default_rng(n_taps)
    coeffs = rng.integers(-1500, 1500, size=n_taps, dtype=np.int64)

    def make():
        e = A.Fir(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_channels=nch, kind="load")
        e.set_coeffs(coeffs)
        return e

    x = torch.from_numpy(rng.integers(-32768, 32768, size=(nk, nch, cs), dtype=np.int16)).cuda()
    _capture_and_check(make, x, (nch, cs), torch.int16)


def test_cic_decimator_calls_replayed_from_a_graph():
    nch, cs, nk, R = 64, 8192, 3, 8
    fin = A.Fmt(32, 16)
    fout = A.Fmt(47, 31)
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31, size=(nk, nch, cs), dtype=np.int64).astype(np.int32)).cuda()
    _capture_and_check(lambda: A.Cic(False, R, 1, 5, fin, fout, n_channels=nch), x, (nch, cs // R), torch.int64)


def test_fused_ddc_calls_replayed_from_a_graph():
    nch, cs, nk = 16, 16 * 256 * 4, 3
    cin, fc, fa, fo = A.Fmt(16, 1), A.Fmt(16, 1), A.Fmt(60, 30), A.Fmt(24, 9, True, "RND", "SAT")
    rng = np.random.default_rng(9)
    c = rng.integers(-3000, 3000, size=127, dtype=np.int64)

    def make():
        d = A.Ddc(16, 1, 5, cin, 127, "SHIFT_REG", fc, fa, fo, n_channels=nch)
        d.set_coeffs(c)
        assert d.path == "fused"
        return d

    x = torch.from_numpy(rng.integers(-32768, 32768, size=(nk, nch, cs), dtype=np.int16)).cuda()
    _capture_and_check(make, x, (nch, cs // 16), torch.int32)


def test_short_fir_calls_flip_the_history_and_replay_in_pairs():
    # 100-sample calls of a 255-tap filter: the new history still holds part of the old one, so the handle flips buffers per
    # call; a pair of calls restores the parity and the graph replays correctly
    nch, cs, nk = 8, 100, 2
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(42, 14), A.Fmt(16, 2, True, "RND", "SAT")
    rng = np.random.default_rng(3)
    coeffs = rng.integers(-1500, 1500, size=255, dtype=np.int64)

    def make():
        e = A.Fir(255, "SHIFT_REG", fin, fc, fa, fo, n_channels=nch, kind="load")
        e.set_coeffs(coeffs)
        return e

    x = torch.from_numpy(rng.integers(-32768, 32768, size=(nk, nch, cs), dtype=np.int16)).cuda()
    _capture_and_check(make, x, (nch, cs), torch.int16, n_replays=4)


def test_calls_whose_phase_would_be_baked_in_are_refused_under_capture():
    """A replay re-runs the host-side phase of capture time: a decimator call that does not consume a multiple of R, and the first
    call of an interpolator (start-up outputs dropped), cannot be replayed and must fail loudly while the stream is capturing."""
    nch = 8
    fin, fout = A.Fmt(32, 16), A.Fmt(47, 31)
    dec = A.Cic(False, 8, 1, 5, fin, fout, n_channels=nch)
    x = torch.zeros((nch, 8192 + 3), dtype=torch.int32, device="cuda")
    dec.run(x[:, :8192])
    it = A.Cic(True, 8, 1, 5, fin, fin, n_channels=1).int_type
    intr = A.Cic(True, 8, 1, 5, fin, A.Fmt(it.W, it.I), n_channels=nch)
    yi = torch.zeros((nch, 8192 * 8 + 64), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    errs = []
    with torch.cuda.graph(g, stream=torch.cuda.Stream()):
        dec.run(x[:, :8192])                      # fine: a multiple of R
        for call in (lambda: dec.run(x[:, :8195]), lambda: intr.run(x[:, :4096], yi)):
            try:
                call()
                errs.append(None)
            except RuntimeError as e:
                errs.append(str(e))
    assert errs[0] and "graph capture" in errs[0], errs
    assert errs[1] and "graph capture" in errs[1], errs


@pytest.mark.parametrize("ifac,n_taps", [(8, 16), (2, 16), (3, 12), (6, 8)])
def test_poly_intr_calls_with_their_side_stream_replayed_from_a_graph(ifac, n_taps):
    """ac_poly_intr forks the head / tail / state kernels of a call onto a stream of its own beside the matrix-core kernel (fir_kernels.hpp:
    SideStream): event record / wait pairs, which a capture turns into graph edges.  The handle flips its state buffers every call, so calls are
    captured in pairs; replays continue the stream.  Factors 2 / 3 / 6 with 2-byte outputs: the head / tail kernels write [0, o_a) and [o_b, n_out)
    of the same rows the matrix-core kernel fills between them, concurrently -- ranges that are closest to sharing a 4-byte word where IF is small."""
    nch, cs, nk = 24, 4096, 2
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
    rng = np.random.default_rng(17)
    csz = (n_taps // 2 - 1) + (ifac - 1) * n_taps // 2 + 1
    c = rng.integers(-3000, 3000, size=csz, dtype=np.int64)
    sign, corr = np.ones(ifac, dtype=np.int64), np.arange(ifac)   # (the bench row's control words: the matrix-core kernel takes them)

    def make():
        e = A.PolyIntr(n_taps, csz, ifac, "FOLD_EVEN", fin, fc, fa, fo, n_channels=nch)
        e.set_ctrl(c, sign, corr)
        return e

    x = torch.from_numpy(rng.integers(-32768, 32768, size=(nk, nch, cs), dtype=np.int16)).cuda()
    n_replays = 3
    # the first call of a stream emits IF - 1 fewer phase groups than every later one: both engines are primed with one eager pair, so that the
    # captured calls have the steady-state output shape; no reset afterwards -- the replays continue the primed stream
    ref_eng = make()
    for k in range(nk):
        ref_eng.run(x[k])
    refs = [torch.stack([ref_eng.run(x[k]).clone() for k in range(nk)]) for _ in range(n_replays)]
    assert ref_eng.path == "mfma_gen", ref_eng.path
    torch.cuda.synchronize()
    eng = make()
    for k in range(nk):
        eng.run(x[k])
    y = torch.zeros_like(refs[0])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.Stream()):
        for k in range(nk):
            y[k].copy_(eng.run(x[k]))
    torch.cuda.synchronize()
    for r in range(n_replays):
        y.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, refs[r]), "replay %d differs from the eager stream" % r
