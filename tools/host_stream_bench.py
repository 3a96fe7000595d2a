#!/usr/bin/env python3
"""tools/host_stream_bench.py -- PCIe-inclusive rate of the config-2 FIR when the samples live in HOST memory (DESIGN 5:
`value` of bench.py is measured with the inputs resident in HBM; this is the other number the boundary owes).

A streaming source hands over chunks of [n_channels][chunk] samples in pinned host buffers.  Three HIP streams overlap the
legs: H2D of chunk k+1, the FIR of chunk k (acdsp_fir_run on the compute stream: the handle carries the filter state from
chunk to chunk, as ac_fir_load_coeffs' shift register does from call to call), D2H of chunk k-1; events order the legs and
recycle the NBUF device buffers.  Checks the streamed result against ONE run() over the whole record on a fresh handle,
then prints one JSON line: overlapped and serial (copy, run, copy) rates, and the bare copy rates.

    python tools/host_stream_bench.py [--channels 1024] [--chunk 65536] [--chunks 16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ac_dsp_amd as A  # noqa: E402
from bench import windowed_sinc_raw  # noqa: E402

NBUF = 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=1024)
    ap.add_argument("--chunk", type=int, default=65536, help="samples per channel and chunk")
    ap.add_argument("--chunks", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    nch, cs, nk = args.channels, args.chunk, args.chunks
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
    coeffs = windowed_sinc_raw(255, 0.1, fc.F)

    def new_engine():
        e = A.Fir(255, "SHIFT_REG", fin, fc, fa, fo, n_channels=nch, kind="load")
        e.set_coeffs(coeffs)
        return e

    rng = np.random.default_rng(7)
    # the source's layout: one [n_channels][chunk] block per chunk, pinned
    h_in = torch.from_numpy(rng.integers(-32768, 32768, size=(nk, nch, cs), dtype=np.int16)).pin_memory()
    h_out = torch.empty((nk, nch, cs), dtype=torch.int16).pin_memory()
    d_in = [torch.empty((nch, cs), dtype=torch.int16, device=dev) for _ in range(NBUF)]
    d_out = [torch.empty((nch, cs), dtype=torch.int16, device=dev) for _ in range(NBUF)]
    s_h2d, s_run, s_d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def streamed(eng):
        ev_in = [torch.cuda.Event() for _ in range(nk)]
        ev_run = [torch.cuda.Event() for _ in range(nk)]
        ev_out = [torch.cuda.Event() for _ in range(nk)]
        for k in range(nk):
            b = k % NBUF
            with torch.cuda.stream(s_h2d):
                if k >= NBUF:
                    s_h2d.wait_event(ev_run[k - NBUF])       # the FIR of the chunk that used this input buffer is done
                d_in[b].copy_(h_in[k], non_blocking=True)
                ev_in[k].record(s_h2d)
            with torch.cuda.stream(s_run):
                s_run.wait_event(ev_in[k])
                if k >= NBUF:
                    s_run.wait_event(ev_out[k - NBUF])       # its output buffer has been drained
                eng.run(d_in[b], out=d_out[b])
                ev_run[k].record(s_run)
            with torch.cuda.stream(s_d2h):
                s_d2h.wait_event(ev_run[k])
                h_out[k].copy_(d_out[b], non_blocking=True)
                ev_out[k].record(s_d2h)
        torch.cuda.synchronize()

    def serial(eng):
        for k in range(nk):
            d_in[0].copy_(h_in[k], non_blocking=True)
            eng.run(d_in[0], out=d_out[0])
            h_out[k].copy_(d_out[0], non_blocking=True)
            torch.cuda.synchronize()

    # parity: streamed chunks (state carried by the handle) == one run over the whole record
    eng = new_engine()
    streamed(eng)
    whole = torch.from_numpy(np.ascontiguousarray(h_in.numpy().transpose(1, 0, 2).reshape(nch, nk * cs))).to(dev)
    ref = new_engine().run(whole).cpu().numpy().reshape(nch, nk, cs).transpose(1, 0, 2)
    assert np.array_equal(ref, h_out.numpy()), "streamed chunks differ from the one-shot run"
    del whole

    def timed(fn):
        best = None
        for _ in range(args.reps):
            e = new_engine()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(e)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    def copy_rate(dst, src):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(nk):
            dst[k % NBUF if isinstance(dst, list) else k].copy_(src[k] if not isinstance(src, list) else src[k % NBUF], non_blocking=True)
        torch.cuda.synchronize()
        return nk * nch * cs * 2 / (time.perf_counter() - t0) / 1e9

    t_over, t_ser = timed(streamed), timed(serial)
    samples = float(nk) * nch * cs
    out = {"workload": "ac_fir_load_coeffs 255-tap <16,2>, %d ch, %d chunks x %d samples from pinned host memory" % (nch, nk, cs),
           "parity": "streamed == one-shot run (bit-exact)",
           "Msamples_per_s_overlapped": samples / t_over / 1e6, "Msamples_per_s_serial": samples / t_ser / 1e6,
           "h2d_GBps": copy_rate(d_in, h_in), "d2h_GBps": copy_rate(h_out, d_out),
           "bytes_per_sample_over_pcie": 4, "pcie_GBps_overlapped_each_way": samples * 2 / t_over / 1e9}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
