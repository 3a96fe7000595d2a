#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X fixed-point streaming-filter engine.

Metric (BASELINE.json): Msamples/s of the 255-tap ac_fixed<16,2> FIR on 1024 channels (configs[1]:
ac_fir_load_coeffs, 1024 channels x 2^20 samples per GPU), inputs resident in HBM.  One "step" = one
pass of the FIR hot path over one [1024][2^20] block of the stream (filter state carries from step to
step, like consecutive run() calls of the reference).  With N GPUs every rank filters its own slice of
N*1024 independent channels: no data-path collective, weak scaling.

Launch: `python bench.py` (1 GPU) or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W`.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 MFMA peak (2x the 2.5 PF bf16 dense peak)


def shard(n_total, world, rank):
    """Contiguous channel slice [lo, hi) of rank `rank` (independent filter objects: no exchange)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def windowed_sinc_raw(n_taps, cutoff, frac_bits):
    m = (n_taps - 1) / 2.0
    k = np.arange(n_taps) - m
    win = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(n_taps) / (n_taps - 1)) if n_taps > 1 else np.ones(1)   # (one tap: unity gain)
    h = np.sinc(2 * cutoff * k) * 2 * cutoff * win
    h = h / h.sum()
    raw = np.round(h * 2.0 ** frac_bits).astype(np.int64)
    return (raw + raw[::-1]) // 2


def cpu_baseline_fir(n_taps, coeffs, fin, fc, fa, fo, seed):
    """Oracle ("port" of the reference's per-sample loop) timed on this host's cores, bounded sample."""
    import threading
    from oracle import OracleFir, Fmt as OFmt, stimulus
    cores = host_cores()
    n = 16384
    of = [OFmt(f.W, f.I, f.S, f.Q, f.O) for f in (fin, fc, fa, fo)]
    xs = stimulus(seed, cores, n, fin.W)
    objs = [OracleFir(n_taps, "SHIFT_REG", *of) for _ in range(cores)]
    state = {"reps": 1}

    def set_reps(r):
        state["reps"] = r

    def work(i):
        for _ in range(state["reps"]):
            objs[i].run(coeffs, xs[i:i + 1])

    reps, dt = timed_threads(work, cores, lambda r: set_reps(r))
    total = cores * n * reps
    return {"value": total / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d channels x %d samples x %d passes of the same 255-tap workload, one oracle object per core "
                      "(ctypes releases the GIL), %.1f s wall" % (cores, n, reps, dt)}


def reference_cfg_coeffs(which, fc):
    """Raw words of the coefficient set of tests/rtest_ac_fir_<which>_coeffs.cpp (tests/golden/ref_txt/ac_fir_<which>_coeffs_cfg.txt: data,
    see NOTICE); every value is exactly representable in its COEFF_TYPE."""
    from fractions import Fraction
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "ref_txt", "ac_fir_%s_coeffs_cfg.txt" % which)
    with open(path) as f:
        vals = [Fraction(t) for t in f.read().replace(",", " ").split()]
    raw = [v * (1 << (fc.W - fc.I)) for v in vals]
    assert all(r.denominator == 1 for r in raw), "cfg value not representable in COEFF_TYPE"
    return np.array([int(r) for r in raw], dtype=np.int64)


def build_workload(workload, args, world, rank, local_rank):
    """Allocate one workload's engine handle and device buffers (per-rank channel slice) and return its description:
    step() = one pass of the hot path over the resident [channels][samples] block."""
    import ac_dsp_amd as A
    dev = torch.device("cuda", local_rank)
    placement = {}

    def shop(shape, dtype, trial):
        """the output buffer of an HBM-bound row: one of --placement K separately allocated candidates, chosen by timing the row's own call on each
        (A.shop_output; INTEGRATION.md 7: the same operator runs up to 12 % apart on different input / output allocation pairs)"""
        kcand = int(getattr(args, "placement", 0) or 0)
        if kcand <= 1:
            return torch.empty(shape, dtype=dtype, device=dev)
        yb, ms = A.shop_output(trial, shape, dtype, dev, candidates=kcand, reps=2)
        placement.update({"candidates": kcand, "trial_ms": [round(t, 4) for t in ms]})
        return yb
    seed = 0xACD5
    coeffs = n_taps = fin = fc = fa = fo = None
    if workload in ("fir255", "fir255_dense", "fir255_wide", "fir1023"):
        n_taps = 1023 if workload == "fir1023" else 255
        ch_per_gpu = args.channels or 1024
        n = args.samples or (1 << 20)
        fin, fc, fa = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12)
        fo = A.Fmt(16, 2, True, "RND", "SAT")
        if workload == "fir1023":     # BASELINE configs[3]: ac_fir_prog_coeffs, 1023 taps, ACC <42,14>, 1024 ch per GPU
            fa = A.Fmt(42, 14)
            coeffs = windowed_sinc_raw(n_taps, 0.05, fc.F)
        elif workload in ("fir255", "fir255_wide"):
            coeffs = windowed_sinc_raw(n_taps, 0.1, fc.F)  # symmetric windowed sinc, sum|c| < 2
            if workload == "fir255_wide":             # SURVEY 8(d) second row: OUT_TYPE = ACC_TYPE, 8-byte containers
                fo = fa
        else:  # every Toeplitz byte-plane block populated
            coeffs = np.random.default_rng(1).integers(-32768, 32640, size=n_taps, dtype=np.int64)
        lo, hi = shard(ch_per_gpu * world, world, rank)
        eng = A.Fir(n_taps, "SHIFT_REG", fin, fc, fa, fo, n_channels=hi - lo, kind="load", device=local_rank)
        eng.set_coeffs(coeffs)
        x = torch.empty((hi - lo, n + args.pad), dtype=torch.int16, device=dev)[:, :n]
        A.fill_stimulus(x, seed, args.stim_bits or 16, ch0=lo)
        y = torch.empty((hi - lo, n + args.pad), dtype=A.torch_dtype_for(fo), device=dev)[:, :n]
        bytes_per_sample = 2.0 + y.element_size()  # 2 B read + 2 B (8 B for the wide row) written
        macs_per_sample = 4.0 * 32 * eng_nb(n_taps)  # int8 MACs of the DENSE 4-byte-plane Toeplitz formulation (32 MAC per MFMA and sample)
        issued_macs_per_sample = 32.0 * eng.mfma_issued()   # ... and of the MFMAs the selected kernel really issues (zero blocks skipped)
        name = "ac_fir_load_coeffs 255-tap ac_fixed<16,2> -> <16,2,RND,SAT>, ACC <40,12>, %d ch x %d samples per GPU " \
               "(BASELINE configs[1])" % (ch_per_gpu, n)
        if workload == "fir255_wide":
            name = name.replace("-> <16,2,RND,SAT>", "-> OUT = ACC <40,12> (int64 containers)").replace("(BASELINE configs[1])", "(BASELINE configs[1], wide-output row)")
        if workload == "fir1023":
            name = "ac_fir_prog_coeffs 1023-tap ac_fixed<16,2> -> <16,2,RND,SAT>, ACC <42,14>, %d ch x %d samples per GPU " \
                   "(BASELINE configs[3])" % (ch_per_gpu, n)
        dtype = "int16 (exact: int8-split MFMA, int32 accumulate)"

        def step():
            eng.run(x, y)
        path = eng.path
        samples_per_step = (hi - lo) * n
    elif workload in ("rtest_const_types", "rtest_load_types", "rtest_prog_types"):
        # The reference's OWN shipped parameterisations (the only FIR configurations with reference-held vectors), at batch scale:
        #   tests/rtest_ac_fir_const_coeffs.cpp:50-74,160   29 taps FOLD_ODD, <16,8> x <32,16> -> ACC = OUT <64,32> (exact sums)
        #   tests/rtest_ac_fir_load_coeffs.cpp:50-74,135    27 taps FOLD_ODD, <32,16> x <32,16> -> <64,32>            (exact sums)
        #   tests/rtest_ac_fir_prog_coeffs.cpp:47-54        27 taps FOLD_ODD, <28,6> x <23,7> -> <64,32>: 6 bits dropped per tap (class B)
        # with the coefficient sets of the testbenches (tests/golden/ref_txt/*_cfg.txt, attributed in NOTICE)
        which = workload.split("_")[1]
        n_taps = 29 if which == "const" else 27
        fin = {"const": A.Fmt(16, 8), "load": A.Fmt(32, 16), "prog": A.Fmt(28, 6)}[which]
        fc = {"const": A.Fmt(32, 16), "load": A.Fmt(32, 16), "prog": A.Fmt(23, 7)}[which]
        fa = fo = A.Fmt(64, 32)
        ch_per_gpu = args.channels or (1024 if which == "const" else 512)
        n = args.samples or (1 << 20)
        coeffs = reference_cfg_coeffs(which, fc)
        assert len(coeffs) == n_taps
        lo, hi = shard(ch_per_gpu * world, world, rank)
        eng = A.Fir(n_taps, "FOLD_ODD", fin, fc, fa, fo, n_channels=hi - lo, kind=which, device=local_rank)
        eng.set_coeffs(coeffs)
        x = torch.empty((hi - lo, n + args.pad), dtype=A.torch_dtype_for(fin), device=dev)[:, :n]
        A.fill_stimulus(x, seed, fin.W, ch0=lo)
        y = torch.empty((hi - lo, n + args.pad), dtype=torch.int64, device=dev)[:, :n]
        bytes_per_sample = float(x.element_size() + 8)
        macs_per_sample = 0.0
        name = "ac_fir_%s_coeffs %d-tap FOLD_ODD <%d,%d> x <%d,%d> -> ACC = OUT <64,32> (the types and coefficients of tests/rtest_ac_fir_%s_coeffs.cpp), " \
               "%d ch x %d samples per GPU" % (which, n_taps, fin.W, fin.I, fc.W, fc.I, which, ch_per_gpu, n)
        dtype = "int64 (exact multi-plane int8 MFMA sums%s)" % (" minus the per-tap dropped bits: class B" if which == "prog" else "")
        coeffs = None

        def step():
            eng.run(x, y)
        path = eng.kernel
        samples_per_step = (hi - lo) * n
    elif workload == "cic_dec_r64":
        # round 6: ac_cic_dec_full at a rate it is deployed at -- R = 64, N = 4 on 16-bit samples (two-stage kernel, ac_dsp_amd/csrc/cic2.hip).
        # ACDSP_BENCH_CIC="W,I,R,M,N" swaps the parameter set (tools/knob_sweep.py: same-process A/B on other rates / widths)
        W, I, R, M, N = (int(v) for v in os.environ.get("ACDSP_BENCH_CIC", "16,1,64,1,4").split(","))
        fin = A.Fmt(W, I)
        ch_per_gpu = args.channels or 4096
        n = args.samples or (1 << 20)
        lo, hi = shard(ch_per_gpu * world, world, rank)
        it = A.Cic(False, R, M, N, fin, fin, n_channels=1, device=local_rank).int_type
        fo = A.Fmt(it.W, it.I)                   # OUT_TYPE = the reference's lossless INT_TYPE
        eng = A.Cic(False, R, M, N, fin, fo, n_channels=hi - lo, device=local_rank)
        x = torch.empty((hi - lo, n), dtype=A.torch_dtype_for(fin), device=dev)
        A.fill_stimulus(x, seed, W, ch0=lo)
        y = shop((hi - lo, (n // R + 8 + 7) // 8 * 8), A.torch_dtype_for(fo), lambda yy: eng.run(x, yy))
        bytes_per_sample = float(x.element_size()) + float(y.element_size()) / R
        macs_per_sample = 0.0
        name = "ac_cic_dec_full N=%d R=%d M=%d ac_fixed<%d,%d> -> <%d,%d>, %d ch x %d samples per GPU" % (N, R, M, W, I, it.W, it.I, ch_per_gpu, n)
        dtype = "int64 (wrap arithmetic mod 2^%d)" % it.W
        coeffs = None

        def step():
            eng.run(x, y)
        samples_per_step = (hi - lo) * n
        step()
        path = "cic_dec_" + eng.path
    elif workload in ("cic_dec_r7m2n4", "cic_intr_r7m2n5"):
        # the reference's CIC testbench parameters (tests/ac_cic_dec_full_param.h:33-47, ac_cic_intr_full_param.h:33-47) at batch scale
        interp = workload.startswith("cic_intr")
        N = 5 if interp else 4
        fin = A.Fmt(32, 16)
        ch_per_gpu = args.channels or (1024 if interp else 4096)
        n = args.samples or ((1 << 18) if interp else 7 * (1 << 17))
        lo, hi = shard(ch_per_gpu * world, world, rank)
        it = A.Cic(interp, 7, 2, N, fin, fin, n_channels=1, device=local_rank).int_type
        fo = A.Fmt(it.W, it.I)                   # <49,33> / <48,32>: the testbenches' OUT_TYPEs
        eng = A.Cic(interp, 7, 2, N, fin, fo, n_channels=hi - lo, device=local_rank)
        x = torch.empty((hi - lo, n), dtype=torch.int32, device=dev)
        A.fill_stimulus(x, seed, 32, ch0=lo)
        y = shop((hi - lo, (n * 7 + 64) if interp else (n // 7 + 8)), torch.int64, lambda yy: eng.run(x, yy))
        if interp:
            eng.run(x[:, :64], y)                # steady state: later calls emit R outputs per input
        bytes_per_sample = 4.0 + (8.0 * 7 if interp else 8.0 / 7)
        macs_per_sample = 0.0
        name = "ac_cic_%s_full N=%d R=7 M=2 ac_fixed<32,16> -> <%d,%d> (the parameters of tests/rtest_ac_cic_%s_full.cpp), %d ch x %d input samples per GPU" % (
            "intr" if interp else "dec", N, it.W, it.I, "intr" if interp else "dec", ch_per_gpu, n)
        dtype = "int64 (wrap arithmetic mod 2^%d)" % it.W
        coeffs = None

        def step():
            eng.run(x, y)
        samples_per_step = (hi - lo) * n
        step()
        path = ("cic_intr_" if interp else "cic_dec_") + eng.path
    elif workload == "polydec":
        # ac_poly_dec, 16 taps per branch x DF = 8 (128-tap decimate-by-8), ac_fixed<16,2>
        ch_per_gpu = args.channels or 1024
        n = args.samples or (1 << 22)
        fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
        lo, hi = shard(ch_per_gpu * world, world, rank)
        eng = A.PolyDec(16, 8, fin, fc, fa, fo, n_channels=hi - lo, device=local_rank)
        hh = windowed_sinc_raw(127, 0.05, fc.F)
        hh = np.concatenate([hh, [0]])
        eng.set_coeffs(np.array([hh[df + tp * 8] for df in range(8) for tp in range(16)], dtype=np.int64))
        x = torch.empty((hi - lo, n + args.pad), dtype=torch.int16, device=dev)[:, :n]
        A.fill_stimulus(x, seed, 16, ch0=lo)
        y = shop((hi - lo, n // 8 + 8), torch.int16, lambda yy: eng.run(x, yy))
        bytes_per_sample = 2.0 + 2.0 / 8
        macs_per_sample = 0.0
        name = "ac_poly_dec NTAPS=16 DF=8 ac_fixed<16,2> -> <16,2,RND,SAT>, %d ch x %d samples per GPU (SURVEY 8 f2)" % (ch_per_gpu, n)
        dtype = "int16 (exact: multi-plane int8 MFMA, 64-bit recombination)"
        coeffs = None

        def step():
            eng.run(x, y)
        samples_per_step = (hi - lo) * n
        path = "polydec"
    elif workload == "polyintr":
        # ac_poly_intr FOLD_EVEN, 16 taps x IF = 8, ac_fixed<16,2>; 8 outputs per input sample
        ch_per_gpu = args.channels or 1024
        n = args.samples or (1 << 18)
        fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
        lo, hi = shard(ch_per_gpu * world, world, rank)
        eng = A.PolyIntr(16, 64, 8, "FOLD_EVEN", fin, fc, fa, fo, n_channels=hi - lo, device=local_rank)
        eng.set_ctrl(windowed_sinc_raw(127, 0.05, fc.F)[:64], [1] * 8, list(range(8)))
        x = torch.empty((hi - lo, n), dtype=torch.int16, device=dev)
        A.fill_stimulus(x, seed, 16, ch0=lo)
        eng.run(x[:, :16])                       # past the stream's first sample: every later call emits IF per input
        bytes_per_sample = 2.0 + 2.0 * 8
        macs_per_sample = 0.0
        name = "ac_poly_intr FOLD_EVEN NTAPS=16 IF=8 ac_fixed<16,2> -> <16,2,RND,SAT>, %d ch x %d input samples per GPU (SURVEY 8 f2)" % (ch_per_gpu, n)
        dtype = "int64 (exact per-MAC order, ACC <40,12>)"
        coeffs = None

        def step():
            eng.run(x)
        samples_per_step = (hi - lo) * n
        step()
        path = "polyintr_" + eng.path
    elif workload == "intgdump":
        # ac_intg_dump, 4 interleaved channels per object, dumps every 64 rounds
        ch_per_gpu = args.channels or 1024
        n = args.samples or (1 << 20)            # interleaved samples per object
        fin, fa, fo = A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16)
        lo, hi = shard(ch_per_gpu * world, world, rank)
        eng = A.IntgDump(64, 4, fin, fa, fo, n_objects=hi - lo, device=local_rank)
        n_sample = np.full(n // (64 * 4), 64, dtype=np.int64)
        x = torch.empty((hi - lo, n), dtype=torch.int16, device=dev)
        A.fill_stimulus(x, seed, 16, ch0=lo)
        bytes_per_sample = 2.0 + 4.0 / 64
        macs_per_sample = 0.0
        name = "ac_intg_dump NS=64 CHN=4 ac_fixed<16,8> -> <32,16>, %d objects x %d interleaved samples per GPU (SURVEY 8 f4)" % (ch_per_gpu, n)
        dtype = "int64 (every add quantised to ACC <32,16>)"
        coeffs = None

        def step():
            eng.run(x, n_sample)
        samples_per_step = (hi - lo) * n
        path = "intgdump"
    elif workload == "mvavg":
        # ac_mv_avg, TAPS = 9, AC_MIRROR, frames of 1024 samples (S_TYPE ac_int<11,false>), ac_fixed<16,8>
        ch_per_gpu = args.channels or 1024
        n = args.samples or (1 << 20)            # samples per object = 1024 frames x 1024
        fin, fc, fa, fo = A.Fmt(16, 8), A.Fmt(16, 2), A.Fmt(40, 18), A.Fmt(16, 8, True, "RND", "SAT")
        lo, hi = shard(ch_per_gpu * world, world, rank)
        eng = A.MvAvg(1024, 9, "MIRROR", fin, fc, fa, fo, n_objects=hi - lo, device=local_rank)
        wts = np.round(np.hanning(11)[1:-1] / np.hanning(11).sum() * 2.0 ** fc.F).astype(np.int64)
        eng.set_coeffs(wts)
        x = torch.empty((hi - lo, n), dtype=torch.int16, device=dev)
        A.fill_stimulus(x, seed, 16, ch0=lo)
        y = shop((hi - lo, n), torch.int16, lambda yy: eng.run(x, 1024, out=yy))
        bytes_per_sample = 2.0 + 2.0
        macs_per_sample = 0.0
        name = "ac_mv_avg TAPS=9 AC_MIRROR ac_fixed<16,8> -> <16,8,RND,SAT>, ACC <40,18>, %d objects x %d frames x 1024 samples per GPU (SURVEY 8 f4)" % (ch_per_gpu, n // 1024)
        dtype = "int32 (exact: weights sum below 2^15, ACC <40,18> cannot wrap)"
        coeffs = None

        def step():
            eng.run(x, 1024, out=y)
        samples_per_step = (hi - lo) * n
        step()
        path = "mvavg_" + eng.path
    elif workload == "cic_intr":
        # ac_cic_intr_full N=5 R=8 on ac_fixed<32,16> (named in north_star; no BASELINE config): 8 outputs per input
        ch_per_gpu = args.channels or 1024
        n = args.samples or (1 << 18)
        fin = A.Fmt(32, 16)
        lo, hi = shard(ch_per_gpu * world, world, rank)
        it = A.Cic(True, 8, 1, 5, fin, fin, n_channels=1, device=local_rank).int_type
        fo = A.Fmt(it.W, it.I)
        eng = A.Cic(True, 8, 1, 5, fin, fo, n_channels=hi - lo, device=local_rank)
        x = torch.empty((hi - lo, n), dtype=torch.int32, device=dev)
        A.fill_stimulus(x, seed, 32, ch0=lo)
        y = shop((hi - lo, n * 8 + 64), torch.int64, lambda yy: eng.run(x, yy))
        eng.run(x[:, :64], y)                    # steady state: later calls emit R outputs per input
        bytes_per_sample = 4.0 + 8.0 * 8
        macs_per_sample = 0.0
        name = "ac_cic_intr_full N=5 R=8 M=1 ac_fixed<32,16> -> <%d,%d>, %d ch x %d input samples per GPU" % (it.W, it.I, ch_per_gpu, n)
        dtype = "int64 (wrap arithmetic mod 2^%d)" % it.W
        coeffs = None

        def step():
            eng.run(x, y)
        samples_per_step = (hi - lo) * n
        step()
        path = "cic_intr_" + eng.path
    elif workload == "ddc":
        # BASELINE configs[4]: CIC R=16 N=5 on ac_fixed<16,1> -> lossless INT <36,21> -> 127-tap FIR (IN <36,21>,
        # COEFF <16,1>); I and Q are separate real streams, 2048 complex = 4096 real streams per GPU
        ch_per_gpu = args.channels or 4096
        n = args.samples or (1 << 20)
        cin, mid = A.Fmt(16, 1), A.Fmt(36, 21)
        fc, fa, fo = A.Fmt(16, 1), A.Fmt(60, 30), A.Fmt(24, 9, True, "RND", "SAT")
        lo, hi = shard(ch_per_gpu * world, world, rank)
        # one handle for the cascade: fused kernel (the 36-bit intermediate never reaches HBM); ACDSP_NO_FUSE=1 -> two kernels
        eng = A.Ddc(16, 1, 5, cin, 127, "SHIFT_REG", fc, fa, fo, n_channels=hi - lo, kind="const", device=local_rank)
        assert (eng.int_type.W, eng.int_type.I) == (mid.W, mid.I)
        coeffs = windowed_sinc_raw(127, 0.2, fc.F)
        eng.set_coeffs(coeffs)
        x = torch.empty((hi - lo, n + args.pad), dtype=torch.int16, device=dev)[:, :n]
        A.fill_stimulus(x, seed, 16, ch0=lo)
        y = shop((hi - lo, n // 16 + 8), torch.int32, lambda yy: eng.run(x, yy))
        bytes_per_sample = 2.0 + 4.0 / 16      # 2 B read per real input sample + 4 B written per 16 (intermediate stays on chip ideally)
        macs_per_sample = 0.0
        name = "DDC: ac_cic_dec_full R=16 N=5 <16,1> -> 127-tap ac_fir_const_coeffs IN <36,21>, %d real streams x %d samples per GPU " \
               "(BASELINE configs[4])" % (ch_per_gpu, n)
        dtype = "int64 (CIC wrap arithmetic, FIR exact dot product)"

        def step():
            eng.run(x, y)
        path = "ddc_" + eng.path
        samples_per_step = (hi - lo) * n
        coeffs = None
    else:
        ch_per_gpu = args.channels or 4096
        n = args.samples or (1 << 22)
        fin, fo = A.Fmt(32, 16), A.Fmt(47, 31)
        lo, hi = shard(ch_per_gpu * world, world, rank)
        eng = A.Cic(False, 8, 1, 5, fin, fo, n_channels=hi - lo, device=local_rank)
        x = torch.empty((hi - lo, n + args.pad), dtype=torch.int32, device=dev)[:, :n]
        A.fill_stimulus(x, seed, 32, ch0=lo)
        y = shop((hi - lo, n // 8 + 8 + args.pad), torch.int64, lambda yy: eng.run(x, yy))
        bytes_per_sample = 5.0                   # 4 B read + 8 B / 8 written
        macs_per_sample = 0.0
        name = "ac_cic_dec_full N=5 R=8 M=1 ac_fixed<32,16> -> <47,31>, %d ch x %d samples per GPU (BASELINE configs[2])" % (ch_per_gpu, n)
        dtype = "int64 (wrap arithmetic mod 2^47)"
        coeffs = None

        def step():
            eng.run(x, y)
        path = "cic_dec"
        samples_per_step = (hi - lo) * n

    if not macs_per_sample:
        issued_macs_per_sample = 0.0
    return {"workload": workload, "name": name, "dtype": dtype, "step": step, "path": path, "samples_per_step": samples_per_step, "placement": placement,
            "bytes_per_sample": bytes_per_sample, "macs_per_sample": macs_per_sample, "issued_macs_per_sample": issued_macs_per_sample, "coeffs": coeffs, "eng": eng, "x": x, "n": n,
            "ch_per_gpu": ch_per_gpu, "n_taps": n_taps, "fmts": (fin, fc, fa, fo), "seed": seed}


def build_node_workload(workload, args, n_shards, one_gpu):
    """--inproc: the same per-GPU workload as build_workload, but ONE process and ONE node-level handle (acdsp_node_*: a contiguous
    channel slice, an engine handle, a stream and a host thread per device; no collective).  Weak scaling: channels per GPU fixed."""
    import ac_dsp_amd as A
    devices = [0] * n_shards if one_gpu else list(range(n_shards))
    seed = 0xACD5

    def fill(bits):
        return lambda t, lo: A.fill_stimulus(t, seed, bits, ch0=lo)
    run_extra = ()
    if workload in ("fir255", "fir255_dense", "fir255_wide", "fir1023"):
        n_taps = 1023 if workload == "fir1023" else 255
        ch, n = args.channels or 1024, args.samples or (1 << 20)
        fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(42, 14) if workload == "fir1023" else A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
        if workload == "fir255_wide":
            fo = fa
        node = A.NodeFir(n_taps, "SHIFT_REG", fin, fc, fa, fo, ch * n_shards, devices, kind="prog" if workload == "fir1023" else "load")
        node.set_coeffs(np.random.default_rng(1).integers(-32768, 32640, size=n_taps, dtype=np.int64) if workload == "fir255_dense"
                        else windowed_sinc_raw(n_taps, 0.05 if workload == "fir1023" else 0.1, fc.F))
        xs, ys = node.alloc(fin, n, fill(16)), node.alloc(fo, n)
        bps = 2.0 + ys[0].element_size()
        name = "ac_fir_%s_coeffs %d-tap ac_fixed<16,2> -> <%d,%d>, %d ch x %d samples per GPU (%s)" % ("prog" if workload == "fir1023" else "load", n_taps, fo.W, fo.I, ch, n, workload)
    elif workload in ("rtest_const_types", "rtest_load_types", "rtest_prog_types"):
        which = workload.split("_")[1]
        n_taps = 29 if which == "const" else 27
        fin = {"const": A.Fmt(16, 8), "load": A.Fmt(32, 16), "prog": A.Fmt(28, 6)}[which]
        fc = {"const": A.Fmt(32, 16), "load": A.Fmt(32, 16), "prog": A.Fmt(23, 7)}[which]
        fa = fo = A.Fmt(64, 32)
        ch, n = args.channels or (1024 if which == "const" else 512), args.samples or (1 << 20)
        node = A.NodeFir(n_taps, "FOLD_ODD", fin, fc, fa, fo, ch * n_shards, devices, kind=which)
        node.set_coeffs(reference_cfg_coeffs(which, fc))
        xs, ys = node.alloc(fin, n, fill(fin.W)), node.alloc(fo, n)
        bps = xs[0].element_size() + 8.0
        name = "ac_fir_%s_coeffs %d-tap FOLD_ODD, the types and coefficients of tests/rtest_ac_fir_%s_coeffs.cpp, %d ch x %d samples per GPU" % (which, n_taps, which, ch, n)
    elif workload in ("cic_dec", "cic_intr", "cic_dec_r7m2n4", "cic_intr_r7m2n5", "cic_dec_r64"):
        interp = "intr" in workload
        R, M, N = (7, 2, 5 if interp else 4) if "r7" in workload else ((64, 1, 4) if "r64" in workload else (8, 1, 5))
        fin = A.Fmt(16, 1) if "r64" in workload else A.Fmt(32, 16)
        ch = args.channels or (1024 if interp else 4096)
        n = args.samples or ((1 << 18) if interp else (7 * (1 << 17) if R == 7 else ((1 << 20) if R == 64 else (1 << 22))))
        it = A.Cic(interp, R, M, N, fin, fin, n_channels=1).int_type
        fo = A.Fmt(it.W, it.I)
        node = A.NodeCic(interp, R, M, N, fin, fo, ch * n_shards, devices)
        xs = node.alloc(fin, n, fill(fin.W))
        ys = node.alloc(fo, (n * R + 64) if interp else (n // R + 8))
        if interp:
            node.run([x[:, :64] for x in xs], ys)      # steady state: later calls emit R outputs per input
        bps = xs[0].element_size() + (8.0 * R if interp else 8.0 / R)
        name = "ac_cic_%s_full N=%d R=%d M=%d ac_fixed<%d,%d> -> <%d,%d>, %d ch x %d input samples per GPU" % ("intr" if interp else "dec", N, R, M, fin.W, fin.I, it.W, it.I, ch, n)
    elif workload == "ddc":
        ch, n = args.channels or 4096, args.samples or (1 << 20)
        cin, fc, fa, fo = A.Fmt(16, 1), A.Fmt(16, 1), A.Fmt(60, 30), A.Fmt(24, 9, True, "RND", "SAT")
        node = A.NodeDdc(16, 1, 5, cin, 127, "SHIFT_REG", fc, fa, fo, ch * n_shards, devices)
        node.set_coeffs(windowed_sinc_raw(127, 0.2, fc.F))
        xs, ys = node.alloc(cin, n, fill(16)), node.alloc(fo, n // 16 + 8)
        bps, name = 2.25, "DDC: ac_cic_dec_full R=16 N=5 <16,1> -> 127-tap ac_fir_const_coeffs, %d real streams x %d samples per GPU" % (ch, n)
    elif workload == "polydec":
        ch, n = args.channels or 1024, args.samples or (1 << 22)
        fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
        node = A.NodePolyDec(16, 8, fin, fc, fa, fo, ch * n_shards, devices)
        hh = np.concatenate([windowed_sinc_raw(127, 0.05, fc.F), [0]])
        node.set_coeffs(np.array([hh[df + tp * 8] for df in range(8) for tp in range(16)], dtype=np.int64))
        xs, ys = node.alloc(fin, n, fill(16)), node.alloc(fo, n // 8 + 8)
        bps, name = 2.25, "ac_poly_dec NTAPS=16 DF=8 ac_fixed<16,2> -> <16,2,RND,SAT>, %d ch x %d samples per GPU" % (ch, n)
    elif workload == "polyintr":
        ch, n = args.channels or 1024, args.samples or (1 << 18)
        fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
        node = A.NodePolyIntr(16, 64, 8, "FOLD_EVEN", fin, fc, fa, fo, ch * n_shards, devices)
        node.set_ctrl(windowed_sinc_raw(127, 0.05, fc.F)[:64], [1] * 8, list(range(8)))
        xs, ys = node.alloc(fin, n, fill(16)), node.alloc(fo, n * 8 + 8)
        node.run([x[:, :16] for x in xs], ys)          # past the stream's first sample
        bps, name = 18.0, "ac_poly_intr FOLD_EVEN NTAPS=16 IF=8 ac_fixed<16,2> -> <16,2,RND,SAT>, %d ch x %d input samples per GPU" % (ch, n)
    elif workload == "intgdump":
        ch, n = args.channels or 1024, args.samples or (1 << 20)
        fin, fa, fo = A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16)
        node = A.NodeIntgDump(64, 4, fin, fa, fo, ch * n_shards, devices)
        xs, ys = node.alloc(fin, n, fill(16)), node.alloc(fo, n // 64 + 8)
        run_extra = (np.full(n // (64 * 4), 64, dtype=np.int64),)
        bps, name = 2.0 + 4.0 / 64, "ac_intg_dump NS=64 CHN=4 ac_fixed<16,8> -> <32,16>, %d objects x %d interleaved samples per GPU" % (ch, n)
    elif workload == "mvavg":
        ch, n = args.channels or 1024, args.samples or (1 << 20)
        fin, fc, fa, fo = A.Fmt(16, 8), A.Fmt(16, 2), A.Fmt(40, 18), A.Fmt(16, 8, True, "RND", "SAT")
        node = A.NodeMvAvg(1024, 9, "MIRROR", fin, fc, fa, fo, ch * n_shards, devices)
        node.set_coeffs(np.round(np.hanning(11)[1:-1] / np.hanning(11).sum() * 2.0 ** fc.F).astype(np.int64))
        xs, ys = node.alloc(fin, n, fill(16)), node.alloc(fo, n)
        run_extra = (1024,)
        bps, name = 4.0, "ac_mv_avg TAPS=9 AC_MIRROR ac_fixed<16,8> -> <16,8,RND,SAT>, %d objects x %d frames x 1024 samples per GPU" % (ch, n // 1024)
    else:
        raise SystemExit("bench.py --inproc: unknown workload %s" % workload)
    for d in set(devices):
        torch.cuda.synchronize(d)
    return {"node": node, "step": lambda: node.run(xs, *run_extra, ys), "name": name, "samples_per_step": ch * n_shards * n, "samples_per_gpu": ch * n,
            "bytes_per_sample": bps, "ch_per_gpu": ch, "n": n, "devices": devices}


def run_inproc(args):
    """`python bench.py --gpus N --inproc`: N shards of one node-level handle in this process.  A step = node.run() = every shard's
    thread launches its slice and waits for it; K steps are timed on the host clock around the blocking calls (the barrier of the
    process-per-GPU contract is the join inside run())."""
    one_gpu = bool(os.environ.get("ACDSP_BENCH_ONE_GPU"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the HIP engine)")
    if not one_gpu and torch.cuda.device_count() < args.gpus:
        raise SystemExit("bench.py: --gpus %d but only %d device(s) visible" % (args.gpus, torch.cuda.device_count()))
    w = build_node_workload(args.workload, args, args.gpus, one_gpu)
    step, node = w["step"], w["node"]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.settle:
        step()
    for _ in range(args.warmup):
        step()
    per = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        per.append(node.last_ms())
    dt = time.perf_counter() - t0
    kmax = sum(m for _, m in per) / len(per)                      # mean over the steps of the slowest shard's kernel time
    # per physical device: a device listed k times (the one-GPU test mode) runs k shards side by side, so its bytes are k shards'
    shards_on_dev = max(w["devices"].count(d) for d in set(w["devices"]))
    nbytes = w["bytes_per_sample"] * w["samples_per_gpu"] * shards_on_dev
    out = {"metric": "Msamples/s", "value": w["samples_per_step"] * args.steps / dt / 1e6, "unit": "Msamples/s", "n_gpus": args.gpus,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "int16 / int32 / int64 (exact)", "data": "synthetic (on-device splitmix64 counter hash, seed 0xACD5)",
           "config": {"workload": w["name"], "channels_per_gpu": w["ch_per_gpu"], "samples_per_step": w["n"], "devices": w["devices"],
                      "parallelism": "channel-slice x%d, no collectives; ONE process, one acdsp_node_* handle (engine handle + stream + host thread per device)" % args.gpus},
           "roofline": {"bound": "hbm", "achieved": nbytes / (kmax * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": nbytes / (kmax * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "basis": "per GPU: algorithmic bytes of the shards on one device / the slowest shard's kernel time (engine handles' HIP events), mean over the steps",
                        "shards_per_device": shards_on_dev,
                        "kernel_ms_per_shard_last_step": per[-1][0]},
           "cpu_baseline": None}
    print(json.dumps(out), flush=True)


def settle_clocks(step, seconds):
    """Untimed pre-conditioning, reported in the JSON line as `clock_settle`: the same step in a loop for `seconds` of wall
    time BEFORE the W warm-up steps.  The part idles at ~100 MHz and its power management needs 20 - 40 ms of this load to
    settle the shader clock (tools/clock_ramp.py: the first 20 steps from idle average 1.42 ms, every later block of 20 steps
    1.00 ms); W = 3 .. 5 warm-up steps end inside that ramp, so without this the K timed steps measure the DVFS transient of
    a cold start instead of the streaming engine.  Nothing inside the timed region changes.  Returns the steps run."""
    n = 0
    if seconds > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            n += 10
    return n


def measure(w, steps, warmup, barrier, settle_s=0.0):
    """Optional clock pre-conditioning (settle_clocks), W untimed steps, then exactly `steps` timed steps bracketed by
    barrier + synchronize.  Returns wall seconds, the dominant kernel's (avg, min) duration from HIP events on the launch
    stream, and the whole-step event time (ms)."""
    step, eng = w["step"], w["eng"]
    w["settle_steps"] = settle_clocks(step, settle_s)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1) / steps
    try:    # shader clock right behind the timed steps (a 40 us spin kernel, outside the timed region): the clock state the row ran in
        import ac_dsp_amd as A
        w["clock_mhz_after"] = round(A.diag_shader_clock_mhz(torch.cuda.current_device()), 1)
    except Exception:   # noqa: BLE001  (a diagnostic: never fails a bench line)
        w["clock_mhz_after"] = None
    if hasattr(eng, "kernel_stats"):
        k_avg, k_min = eng.kernel_stats(min(steps, 64))   # HIP events around the dominant kernel, on the stream it is launched on
    else:                                                # handles without a kernel timer: events around the whole step
        k_avg = k_min = ev_ms
    return dt, k_avg, k_min, ev_ms


PROFILE_TAGS = {w_: w_ for w_ in ("fir255", "fir255_dense", "fir255_wide", "fir1023", "cic_dec", "ddc", "polydec", "cic_intr", "polyintr", "intgdump", "mvavg",
                                   "rtest_const_types", "rtest_load_types", "rtest_prog_types", "cic_dec_r7m2n4", "cic_intr_r7m2n5", "cic_dec_r64")}


def roofline_of(w, k_avg, k_min, ev_ms):
    """roofline object of one workload: algorithmic bytes per launch / the dominant kernel's average duration (kernel only:
    the per-step state-update kernel and any staging copy are in `frac_step`, which divides by the whole step's event time)."""
    nbytes = w["bytes_per_sample"] * w["samples_per_step"]
    ach = nbytes / (k_avg * 1e-3) / 1e9
    xb = w["x"].element_size()
    traffic, src = pmc_traffic(PROFILE_TAGS.get(w["workload"], "none"))
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "basis": "dominant kernel only, HIP events on its launch stream",
            "frac_step": nbytes / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src,
            "algorithmic_bytes_per_launch": nbytes, "kernel_ms_avg": k_avg, "kernel_ms_min": k_min,
            "algorithmic_bytes_per_sample": w["bytes_per_sample"],
            # north_star's wording "HBM-read roofline": input bytes only (SURVEY 8d asks for both views)
            "read_only": {"bytes_per_sample": float(xb), "achieved": xb * w["samples_per_step"] / (k_avg * 1e-3) / 1e9,
                          "frac": xb * w["samples_per_step"] / (k_avg * 1e-3) / 1e9 / HBM_PEAK_GBS}}


def diag_of(w, k_avg, roof):
    """Driver-timed reference speeds beside a FIR row's roofline, measured HERE -- same process, same input buffer, clocks settled by
    the workload that just ran (acdsp_diag_* in include/acdsp.h, kernels in ac_dsp_amd/csrc/diag.hip):
      copy_GBps         plain 16-byte-per-thread device copy of the workload's input block ("also report vs a measured
                        device-copy bandwidth")
      envelope_ms       2 B in + 2 B out per sample streamed in the best geometry the part offers while issuing the SAME number of
                        int8 MFMAs per 1024 samples as the product kernel (same split between low- and high-byte-plane fragments of
                        the same coefficient set) and nothing else: no byte-plane split, no LDS, no epilogue
      frac_of_envelope  envelope_ms / kernel_ms_avg: 1.0 = the kernel runs at the envelope of its own exact formulation."""
    import ac_dsp_amd as A
    x = w["x"]
    if not x.is_contiguous():
        return
    scratch = torch.empty_like(x)
    cms, cbytes = A.diag_copy_ms(x, scratch, warmup=10, reps=20)
    roof["copy_GBps"] = 2.0 * cbytes / (cms * 1e-3) / 1e9
    roof["frac_of_copy"] = roof["achieved"] / roof["copy_GBps"]
    if w["issued_macs_per_sample"] and w["coeffs"] is not None and w["x"].element_size() == 2 and w["bytes_per_sample"] == 4:
        issued = int(round(w["issued_macs_per_sample"] / 32.0))
        hi = issued - 2 * eng_nb_plan(w["n_taps"])
        try:
            ems, ebytes = A.diag_fir_envelope_ms(w["coeffs"], issued, hi, x, scratch, warmup=20, reps=20)
            sms, _ = A.diag_fir_envelope_ms(None, 0, 0, x, scratch, warmup=10, reps=20)
            # the envelope moves 2 B + 2 B per sample whatever the product's output width: scale nothing, compare times per launch
            roof["envelope_ms"] = ems
            roof["envelope_stream_only_ms"] = sms
            roof["envelope_mfma_per_1024_samples"] = {"issued": issued, "high_plane": hi}
            roof["frac_of_envelope"] = ems / k_avg
            # round 6: the same MFMA counts in the plain copy's geometry (one 16-byte element per thread): is the 32 KB-span stream leg the best one?
            gms, _ = A.diag_fir_envelope_copygeom_ms(w["coeffs"], issued, hi, x, scratch, warmup=20, reps=20)     # four elements per thread, the set's fragments
            gs1, _ = A.diag_fir_envelope_copygeom_ms(None, issued, hi, x, scratch, warmup=20, reps=20)           # one element per thread, stand-in A operands
            g0, _ = A.diag_fir_envelope_copygeom_ms(None, 0, 0, x, scratch, warmup=10, reps=20)
            # ... the same kernel and MFMA count on the fragments of a set with ONE non-zero tap (all-zero operands but one row), and without MFMAs
            zc = np.zeros(len(w["coeffs"]), dtype=np.int64)
            zc[len(zc) // 2] = 1
            gz, _ = A.diag_fir_envelope_copygeom_ms(zc, issued, hi, x, scratch, warmup=20, reps=20)
            g4, _ = A.diag_fir_envelope_copygeom_ms(zc, 0, 0, x, scratch, warmup=10, reps=20)
            roof["envelope_copy_geometry_ms"] = gms
            roof["envelope_copy_geometry_zero_fragments_ms"] = gz
            roof["envelope_copy_geometry_4_per_thread_stream_only_ms"] = g4
            roof["envelope_copy_geometry_standin_ms"] = gs1
            roof["envelope_copy_geometry_stream_only_ms"] = g0
            roof["envelope_copy_geometry_note"] = ("the envelope's MFMA count in the plain copy's geometry (256-thread workgroups in memory order, four 16-byte elements per thread, "
                                                   "the set's Toeplitz fragments loaded once per wave); _zero_fragments: the same instructions on the fragments of a one-tap set; "
                                                   "_standin: one element per thread, A operand = the loaded bytes rotated (every MFMA of a wave repeats the same operands); the "
                                                   "MFMAs cost time only where their operands toggle: the package power cap, not issue or geometry (DESIGN 5.2)")
            roof["envelope_note"] = ("stream (2 B in + 2 B out per sample, 32 KB spans, 8-load / 8-store non-temporal bursts) + the product's MFMA count on "
                                     "Toeplitz fragments of the same coefficient set, nothing else in the loop; timed in this process after the workload "
                                     "(ac_dsp_amd/csrc/diag.hip)")
        except A.AcdspError as e:
            roof["envelope_error"] = str(e)
    del scratch


def eng_nb_plan(n_taps):
    """K-blocks of the MFMA plan (fir_mfma_plan_blocks): even counts of 10 .. 32 blocks are padded to the next odd one."""
    nb = (n_taps - 1 + 31) // 32 + 1
    return nb + 1 if 10 <= nb <= 32 and nb % 2 == 0 else nb


def mfma_roofline_of(w, k_avg):
    """Matrix-pipe utilisation from the MFMAs the kernel ISSUES (acdsp_fir_mfma_issued: all-zero high-byte Toeplitz blocks are
    skipped); the figure of the dense 4-plane formulation is kept beside it, labelled."""
    tops = 2.0 * w["issued_macs_per_sample"] * w["samples_per_step"] / (k_avg * 1e-3) / 1e12
    dense = 2.0 * w["macs_per_sample"] * w["samples_per_step"] / (k_avg * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": tops, "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s (int8 ops of the 32x32x32 MFMAs issued)",
            "frac": tops / I8_MFMA_PEAK_TOPS, "mfma_per_1024_samples": w["issued_macs_per_sample"] / 32.0,
            "dense_formulation": {"mfma_per_1024_samples": w["macs_per_sample"] / 32.0, "achieved": dense, "frac": dense / I8_MFMA_PEAK_TOPS}}


# every other workload, measured in the same process after the headline (N = 1 only): the other BASELINE configurations
# first, then the SURVEY 8 (f) rows
SECONDARY = ["fir255_dense", "fir255_wide", "fir1023", "cic_dec", "ddc", "cic_intr", "polydec", "polyintr", "intgdump", "mvavg",
             # round 5: the reference's own shipped testbench parameterisations at batch scale
             "rtest_const_types", "rtest_load_types", "rtest_prog_types", "cic_dec_r7m2n4", "cic_intr_r7m2n5",
             # round 6: CIC decimation at R >= 32 (two-stage kernel)
             "cic_dec_r64"]
ALL_WORKLOADS = ["fir255"] + SECONDARY


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="fir255", choices=ALL_WORKLOADS)
    ap.add_argument("--channels", type=int, default=0, help="channels per GPU (default: the BASELINE config)")
    ap.add_argument("--samples", type=int, default=0, help="samples per channel per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE configs after the headline workload")
    ap.add_argument("--pad", type=int, default=0, help="extra elements per row (row stride = samples + pad)")
    ap.add_argument("--placement", type=int, default=6, help="output buffers of the HBM-bound decimator rows: candidates to allocate and time the row's own call on "
                    "(the fastest is kept, outside the timed region; 0 / 1: plain allocation).  INTEGRATION.md 7")
    ap.add_argument("--stim-bits", type=int, default=0, help="diagnostic: amplitude of the FIR stimulus in bits (default: full 16)")
    ap.add_argument("--settle", type=float, default=0.3, help="seconds of untimed pre-conditioning steps in front of the warm-up "
                    "(shader-clock ramp from idle; 0 = cold start, the K timed steps then include the DVFS transient)")
    ap.add_argument("--inproc", action="store_true", help="one process, one node-level handle (acdsp_node_*) with --gpus shards, instead "
                    "of one process per GPU")
    args = ap.parse_args()

    if args.inproc:
        return run_inproc(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the HIP engine)")
    if os.environ.get("ACDSP_BENCH_ONE_GPU"):      # self-test of the multi-rank path on a 1-GPU box: every rank on device 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        # Channels are independent filter objects: the data path has no exchange step, so no RCCL communicator is built.
        # The timing contract (barrier, MAX of the times, SUM of the samples) runs over gloo on host scalars.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo announces its connections on the C-level stdout ("[Gloo] Rank 0 is connected to ..."): keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (one process per GPU; without a launcher bench.py spawns the ranks itself)" % (args.gpus, world))
    if not os.environ.get("ACDSP_BENCH_ONE_GPU") and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d device(s) visible" % (world, torch.cuda.device_count()))

    w = build_workload(args.workload, args, world, rank, local_rank)

    def barrier():
        if world > 1:
            dist.barrier()

    cold = None
    if args.settle > 0 and world == 1:
        # the same W + K steps from an idle GPU first (what `--settle 0` prints as the headline): the DVFS transient of a cold start
        torch.cuda.synchronize()
        time.sleep(1.0)
        cdt, ck, _, _ = measure(w, args.steps, args.warmup, barrier, 0.0)
        cold = {"ms_per_step": cdt / args.steps * 1e3, "value": w["samples_per_step"] * args.steps / cdt / 1e6, "kernel_ms_avg": ck,
                "roofline_frac": w["bytes_per_sample"] * w["samples_per_step"] / (ck * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "same W warm-up + K timed steps started 1 s after the GPU went idle, no pre-conditioning"}
    dt, k_avg, k_min, ev_ms = measure(w, args.steps, args.warmup, barrier, args.settle)
    samples_per_step = w["samples_per_step"]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([float(samples_per_step)], dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_samples_per_step = float(tot.item())
    else:
        total_samples_per_step = float(samples_per_step)

    if rank == 0:
        value = total_samples_per_step * args.steps / dt / 1e6
        out = {
            "metric": "Msamples/s", "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": w["dtype"], "data": "synthetic (on-device splitmix64 counter hash, seed 0xACD5)",
            "config": {"workload": w["name"], "kernel_path": w["path"], "channels_per_gpu": w["ch_per_gpu"], "samples_per_step": w["n"],
                       "parallelism": "channel-slice x%d, no collectives" % world},
            "roofline": roofline_of(w, k_avg, k_min, ev_ms),
            "event_ms_per_step": ev_ms,
            "clock_mhz_after": w.get("clock_mhz_after"),
            "clock_settle": {"seconds": args.settle, "steps": w["settle_steps"],
                             "note": "untimed pre-conditioning in front of the W warm-up steps (same step, same data): the shader clock needs "
                                     "20-40 ms of load to settle from idle; --settle 0 measures the cold-start transient instead"},
        }
        if world == 1:
            diag_of(w, k_avg, out["roofline"])
        if w.get("placement"):
            out["config"]["placement"] = w["placement"]
        if cold is not None:
            out["cold_start"] = cold
        if world == 1 and args.workload == "fir255" and not args.no_secondary:
            # what a fresh process pays before its first output (one code object with every kernel): tools/load_cost.py as a child process
            try:
                import subprocess
                lc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "load_cost.py")], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120)
                out.setdefault("cold_start", {})["load_ms"] = json.loads(lc.stdout.decode().strip().splitlines()[-1])
            except Exception as e:  # noqa: BLE001
                out.setdefault("cold_start", {})["load_ms_error"] = str(e)[:200]
        if w["macs_per_sample"]:
            out["mfma_roofline"] = mfma_roofline_of(w, k_avg)
        fin, fc, fa, fo = w["fmts"]
        cpu_args = (w["n_taps"], w["coeffs"], fin, fc, fa, fo, w["seed"])
        if world == 1 and not args.no_secondary and args.workload == "fir255" and not (args.channels or args.samples or args.stim_bits):
            # every other workload in the same process, 10 timed steps each behind 5 warm-up steps (one resident workload at a
            # time: config 3 alone holds 86 GB); measured BEFORE the CPU baseline, whose 10 s leave the GPU idle and in a low
            # power state (the first workload behind it measured 15 % slow)
            del w
            torch.cuda.empty_cache()
            sec = {}
            for name in SECONDARY:
                w2 = build_workload(name, args, 1, 0, local_rank)
                dt2, ka2, km2, ev2 = measure(w2, 10, 5, lambda: None, args.settle)
                r2 = roofline_of(w2, ka2, km2, ev2)
                sec[name] = {"workload": w2["name"], "kernel_path": w2["path"], "ms_per_step": dt2 / 10 * 1e3,
                             "Msamples_per_s": w2["samples_per_step"] * 10 / dt2 / 1e6, "kernel_ms_avg": ka2,
                             "roofline_frac": r2["frac"], "roofline_frac_step": r2["frac_step"], "achieved_GBps": r2["achieved"],
                             "traffic": r2.get("traffic"), "clock_mhz_after": w2.get("clock_mhz_after")}
                if w2.get("placement"):
                    sec[name]["placement"] = w2["placement"]
                if w2["macs_per_sample"]:
                    m2 = mfma_roofline_of(w2, ka2)
                    sec[name]["mfma_frac"] = m2["frac"]
                    sec[name]["mfma_per_1024_samples"] = m2["mfma_per_1024_samples"]
                    if name in ("fir255_dense", "fir1023"):      # int16 in / int16 out rows: the envelope moves the same bytes
                        diag_of(w2, ka2, r2)
                        for k in ("copy_GBps", "envelope_ms", "envelope_stream_only_ms", "frac_of_envelope", "envelope_mfma_per_1024_samples", "envelope_error",
                                  "envelope_copy_geometry_ms", "envelope_copy_geometry_zero_fragments_ms", "envelope_copy_geometry_4_per_thread_stream_only_ms",
                                  "envelope_copy_geometry_standin_ms", "envelope_copy_geometry_stream_only_ms"):
                            if k in r2:
                                sec[name][k] = r2[k]
                del w2
                torch.cuda.empty_cache()
            out["secondary"] = sec
        if world == 1 and not args.no_cpu_baseline and cpu_args[1] is not None:
            out["cpu_baseline"] = cpu_baseline_fir(*cpu_args)
            if args.workload in ("fir255", "fir255_dense"):
                ref = cpu_baseline_ref_headers()
                if ref is not None:
                    out["cpu_baseline_ref_headers"] = ref
        elif world == 1 and not args.no_cpu_baseline and args.workload == "cic_dec":
            out["cpu_baseline"] = cpu_baseline_cic(fin, fo, cpu_args[6])
        if world == 1 and not args.no_cpu_baseline and args.workload == "fir255":
            out["cfg1_cpu"] = cpu_cfg1()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` with no launcher (no WORLD_SIZE in the environment): spawn the N ranks -- one process per GPU,
    the same environment a `torch.distributed.run --nproc-per-node N` launch gives them -- and wait.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = pr.wait() or rc
    return rc


def cpu_baseline_ref_headers():
    """The reference's own header-only loop (ac_fir_load_coeffs::run -> firShiftReg + MAC, /root/reference/include/ac_dsp/
    ac_fir_load_coeffs.h:180-188,320-365) timed on this node's host cores: tests/_bin/cref_bench is prebuilt by
    __graft_entry__.build() in the build container from tools/gen_golden/cref_bench.cpp (the reference's headers compiled where they
    lie over this repo's include/ac_types subset -- hlslibs/ac_types is not in the image, so this times the reference's loop
    structure and ac_channel traffic, not Siemens' ac_int arithmetic).  Only the binary travels.  None when it is missing."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "_bin", "cref_bench")
    if not os.path.isfile(exe):
        return None
    cores = host_cores()
    try:
        n = 65536
        out = subprocess.run([exe, str(cores), str(n), "fir"], capture_output=True, text=True, timeout=120).stdout
        f = [l for l in out.splitlines() if l.startswith("CREF fir")][0].split()
        if float(f[5]) < 2.0:          # scale the sample to ~8 s of wall time
            n = int(min(4 << 20, n * 8.0 / max(float(f[5]), 1e-3)))
            out = subprocess.run([exe, str(cores), str(n), "fir"], capture_output=True, text=True, timeout=300).stdout
            f = [l for l in out.splitlines() if l.startswith("CREF fir")][0].split()
        return {"value": float(f[2]), "unit": "Msamples/s", "cores": int(f[3]), "kind": "reference-headers-over-own-ac_types",
                "sample": "%s filter objects (one per thread) x %s samples of the 255-tap ac_fir_load_coeffs workload (IN/COEFF <16,2>, ACC <40,12>, "
                          "OUT <16,2,RND,SAT>, SHIFT_REG), %s s wall; g++ -O2 -std=c++11" % (f[3], f[4], f[5])}
    except (OSError, IndexError, ValueError, subprocess.SubprocessError) as e:
        return {"error": repr(e), "kind": "reference-headers-over-own-ac_types"}


def cpu_cfg1():
    """BASELINE configs[0] (the one CPU-only configuration) exactly as SURVEY 8(d) fixes it: ac_fir_const_coeffs, 63 taps <16,2,true>,
    ACC = OUT <38,10>, ONE channel, 1024 samples of the reference testbench's two-tone stimulus.  Inputs, coefficients and expected
    outputs come from tests/golden/ref_hdr/fir_cfg1_63.json (dumped from the reference's own header, tools/gen_golden/gen_cfg1.cpp);
    the oracle port is checked bit for bit against them while it is timed (one core: one channel), and the prebuilt reference-header
    binary runs the same configuration (`cref_bench 1 N cfg1`)."""
    import subprocess
    from oracle import OracleFir, Fmt as OFmt
    out = {"config": "ac_fir_const_coeffs 63-tap ac_fixed<16,2,true>, ACC = OUT <38,10>, FOLD_ODD, 1 channel x 1024 samples, two-tone stimulus (BASELINE configs[0])"}
    path = os.path.join(ROOT, "tests", "golden", "ref_hdr", "fir_cfg1_63.json")
    try:
        case = [c for c in json.load(open(path))["cases"] if c["ftype"] == "FOLD_ODD"][0]
        f = [OFmt(a[0], a[1], bool(a[2]), a[3], a[4]) for a in (case["in"], case["coeff"], case["acc"], case["out"])]
        x = np.array(case["x"], dtype=np.int64)[None, :]
        c = np.array(case["coeffs"], dtype=np.int64)
        want = np.array(case["y"], dtype=np.int64)
        xt = np.tile(x, 64)                                     # the record 64 times over per call (ctypes call overhead out of the way)
        reps, t0, ok = 0, time.perf_counter(), True
        while time.perf_counter() - t0 < 2.0:
            y = OracleFir(case["n_taps"], "FOLD_ODD", *f).run(c, xt)[0]
            ok = ok and np.array_equal(y[:want.size], want)     # a fresh object's first 1024 outputs are the fixture's
            reps += 1
        dt = time.perf_counter() - t0
        out["oracle_port"] = {"value": reps * xt.shape[1] / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
                              "bit_exact_vs_reference_header_vectors": bool(ok),
                              "sample": "%d calls of 64 x the 1024-sample record (fresh filter object each), %.1f s wall" % (reps, dt)}
    except (OSError, KeyError, IndexError, ValueError) as e:
        out["oracle_port"] = {"error": repr(e)}
    exe = os.path.join(ROOT, "tests", "_bin", "cref_bench")
    if os.path.isfile(exe):
        try:
            txt = subprocess.run([exe, "1", str(1 << 21), "cfg1"], capture_output=True, text=True, timeout=120).stdout
            fl = [l for l in txt.splitlines() if l.startswith("CREF cfg1")][0].split()
            out["ref_headers"] = {"value": float(fl[2]), "unit": "Msamples/s", "cores": 1, "kind": "reference-headers-over-own-ac_types",
                                  "sample": "%s samples in run() calls of 1024, %s s inside run(); g++ -O2 -std=c++11" % (fl[4], fl[5])}
        except (OSError, IndexError, ValueError, subprocess.SubprocessError) as e:
            out["ref_headers"] = {"error": repr(e)}
    return out


def host_cores():
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota)."""
    try:
        n = max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def pmc_traffic(tag):
    """(HBM bytes per launch of the dominant kernel, source) from the committed rocprofv3 --pmc summary of the same command
    (profiles/r<round>_<tag>_rocprof.txt, newest round first): FETCH_SIZE (KB, x2 on gfx950 for 16-byte streaming reads, see
    MI355X_MICROARCH.md) + WRITE_SIZE (KB).  A static value from the profile of the same binary, NOT measured in this run
    (the PMC passes are separate rocprofv3 invocations); (None, None) when no summary has been committed."""
    for rnd in ("r6", "r5", "r4", "r3", "r2", "r1"):
        path = os.path.join(ROOT, "profiles", "%s_%s_rocprof.txt" % (rnd, tag))
        try:
            vals = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
            seen = set()
            kernel = None
            for line in open(path):
                if line.startswith("void ") or line.startswith("acdsp::"):
                    kernel = line.strip()
                f = line.split()
                if len(f) == 3 and f[0] in vals and kernel and any(k in kernel for k in ("fir_mfma", "fir_gen", "cascade_kernel", "cic_kernel<", "fir_up_kernel", "intg_dump_stream", "intg_dump_batch", "mv_avg_stream")):
                    vals[f[0]] += float(f[2])      # summed over the data-path kernels of one step (the DDC has two)
                    seen.add(f[0])
            if len(seen) == 2:
                return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, "static: profiles/%s_%s_rocprof.txt (separate rocprofv3 --pmc passes of this command)" % (rnd, tag)
        except OSError:
            pass
    return None, None


def timed_threads(work, cores, set_reps):
    """Run work(i) on `cores` threads: one all-core calibration pass, then ~10 s of wall time."""
    import threading

    def go():
        th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0
    set_reps(1)
    t1 = go()
    reps = max(1, min(1000, int(10.0 / max(t1, 1e-3))))
    set_reps(reps)
    return reps, go()


def eng_nb(n_taps):
    return (n_taps - 1 + 31) // 32 + 1


def cpu_baseline_cic(fin, fo, seed):
    from oracle import OracleCic, Fmt as OFmt, stimulus
    cores = host_cores()
    n = 1 << 18
    xs = stimulus(seed, cores, n, fin.W)
    objs = [OracleCic(0, 8, 1, 5, OFmt(fin.W, fin.I, fin.S, fin.Q, fin.O), OFmt(fo.W, fo.I, fo.S, fo.Q, fo.O)) for _ in range(cores)]
    state = {"reps": 1}

    def work(i):
        for _ in range(state["reps"]):
            objs[i].run(xs[i:i + 1])
    reps, dt = timed_threads(work, cores, lambda r: state.__setitem__("reps", r))
    return {"value": cores * n * reps / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d channels x %d samples x %d passes, one oracle object per core, %.1f s wall" % (cores, n, reps, dt)}


if __name__ == "__main__":
    main()
