"""GPU tests of the streaming usage: chunks arriving from pinned host memory, copies and the FIR / CIC on separate HIP
streams ordered by events (tools/host_stream_bench.py is the measured version).  The handle carries the filter state from
chunk to chunk exactly as the reference objects carry their shift registers from call to call
(ac_fir_load_coeffs.h:180-188, ac_cic_full_core.h:71-74), so the streamed result must equal one run over the record."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from oracle import OracleFir
from helpers import ofmt

pytestmark = pytest.mark.gpu


def _pipeline(run, h_in, h_out, d_in, d_out):
    nk, nbuf = h_in.shape[0], len(d_in)
    dev = d_in[0].device
    s_h2d, s_run, s_d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ev_in = [torch.cuda.Event() for _ in range(nk)]
    ev_run = [torch.cuda.Event() for _ in range(nk)]
    ev_out = [torch.cuda.Event() for _ in range(nk)]
    for k in range(nk):
        b = k % nbuf
        with torch.cuda.stream(s_h2d):
            if k >= nbuf:
                s_h2d.wait_event(ev_run[k - nbuf])
            d_in[b].copy_(h_in[k], non_blocking=True)
            ev_in[k].record(s_h2d)
        with torch.cuda.stream(s_run):
            s_run.wait_event(ev_in[k])
            if k >= nbuf:
                s_run.wait_event(ev_out[k - nbuf])
            run(d_in[b], d_out[b])
            ev_run[k].record(s_run)
        with torch.cuda.stream(s_d2h):
            s_d2h.wait_event(ev_run[k])
            h_out[k].copy_(d_out[b], non_blocking=True)
            ev_out[k].record(s_d2h)
    torch.cuda.synchronize()


@pytest.mark.parametrize("n_taps,cs", [(255, 4096), (63, 1000), (1023, 2048)])
def test_fir_chunks_on_three_streams_equal_one_run_and_oracle(n_taps, cs):
    nch, nk = 24, 7
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(42, 14), A.Fmt(16, 2, True, "RND", "SAT")
    rng = np.random.default_rng(n_taps)
    coeffs = rng.integers(-2000, 2000, size=n_taps, dtype=np.int64)

    def engine():
        e = A.Fir(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_channels=nch, kind="load")
        e.set_coeffs(coeffs)
        return e

    x = rng.integers(-32768, 32768, size=(nk, nch, cs), dtype=np.int16)
    h_in = torch.from_numpy(x).pin_memory()
    h_out = torch.empty((nk, nch, cs), dtype=torch.int16).pin_memory()
    d_in = [torch.empty((nch, cs), dtype=torch.int16, device="cuda") for _ in range(2)]
    d_out = [torch.empty((nch, cs), dtype=torch.int16, device="cuda") for _ in range(2)]
    eng = engine()
    _pipeline(lambda a, o: eng.run(a, out=o), h_in, h_out, d_in, d_out)
    whole = np.ascontiguousarray(x.transpose(1, 0, 2).reshape(nch, nk * cs))
    ref = engine().run(torch.from_numpy(whole).cuda()).cpu().numpy()
    got = h_out.numpy().transpose(1, 0, 2).reshape(nch, nk * cs)
    assert np.array_equal(got, ref)
    # two channels against the oracle
    for c in (0, nch - 1):
        orc = OracleFir(n_taps, "SHIFT_REG", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=1)
        yo = orc.run(coeffs, whole[c:c + 1].astype(np.int64))
        assert np.array_equal(got[c].astype(np.int64), np.asarray(yo).reshape(-1))
