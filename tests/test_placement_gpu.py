"""Placement helpers of the C ABI (include/acdsp.h: acdsp_dev_alloc_paired, acdsp_dev_alloc_shop, acdsp_diag_mix_ms) and their torch twins
(A.empty_paired, A.shop_output): the HBM-bound operators run up to 12 % apart on different (input, output) allocation pairs
(profiles/r6_placement.txt), so a caller may allocate a few candidates and keep the fastest.  These tests check the mechanics -- a usable block
comes back, the probe / trial ran on every candidate, the losers were freed, results through the chosen block are the oracle's -- not the speed."""
import ctypes as C

import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from ac_dsp_amd._lib import lib
from helpers import ofmt
from oracle import OracleCic

pytestmark = pytest.mark.gpu


def free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def test_dev_alloc_paired_probes_every_candidate_and_frees_the_losers():
    x = torch.zeros((64, 1 << 18), dtype=torch.int16, device="cuda")          # 32 MB partner
    before = free_bytes()
    ptr, ms = C.c_void_p(), (C.c_float * 5)()
    nbytes = 4 << 20
    assert lib.acdsp_dev_alloc_paired(0, nbytes, C.c_void_p(x.data_ptr()), x.numel() * 2, 1, 5, C.byref(ptr), ms) == 0
    assert ptr.value and all(t > 0 for t in ms)
    assert before - free_bytes() < 3 * nbytes                                    # one block kept (allocator granularity aside), four freed
    assert lib.acdsp_dev_free(0, ptr) == 0
    # no partner: a plain allocation
    assert lib.acdsp_dev_alloc_paired(0, nbytes, None, 0, 1, 5, C.byref(ptr), None) == 0 and ptr.value
    assert lib.acdsp_dev_free(0, ptr) == 0


def test_dev_alloc_shop_runs_the_callers_own_call_on_every_candidate():
    fin, R, M, N = A.Fmt(16, 1), 64, 1, 3
    it = A.Cic(False, R, M, N, fin, fin).int_type
    fo = A.Fmt(it.W, it.I)
    n_ch, n = 8, 1 << 16
    rng = np.random.default_rng(5)
    xh = rng.integers(-32768, 32768, size=(n_ch, n), dtype=np.int64)
    x = torch.from_numpy(xh).to(torch.int16).cuda()
    probe = A.Cic(False, R, M, N, fin, fo, n_channels=n_ch)                       # the trial runs on its own handle: the stream's state stays put
    rows = n // R + 8
    seen = []
    TRIAL = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p)

    def trial(ctx, cand):
        seen.append(cand)
        n_out = C.c_int64()
        return lib.acdsp_cic_run(probe._h, C.c_void_p(x.data_ptr()), x.stride(0), n, C.c_void_p(cand), rows, C.byref(n_out), None)

    ptr, ms = C.c_void_p(), (C.c_float * 3)()
    cb = TRIAL(trial)
    assert lib.acdsp_dev_alloc_shop(0, n_ch * rows * 8, 3, cb, None, 2, C.byref(ptr), ms) == 0
    assert ptr.value and len(set(seen)) == 3 and len(seen) == 3 * 2 * 3 and all(t > 0 for t in ms)
    # the stream itself, through the chosen block
    cic = A.Cic(False, R, M, N, fin, fo, n_channels=n_ch)
    n_out = C.c_int64()
    assert lib.acdsp_cic_run(cic._h, C.c_void_p(x.data_ptr()), x.stride(0), n, ptr, rows, C.byref(n_out), None) == 0
    torch.cuda.synchronize()
    y = np.empty((n_ch, rows), dtype=np.int64)
    assert lib.acdsp_copy_d2h(0, y.ctypes.data_as(C.c_void_p), ptr, y.nbytes) == 0
    yo = OracleCic(False, R, M, N, ofmt(fin), ofmt(fo), n_ch=n_ch).run(xh)
    assert np.array_equal(y[:, :n_out.value], yo)
    assert lib.acdsp_dev_free(0, ptr) == 0


def test_torch_twins():
    x = torch.zeros((64, 1 << 18), dtype=torch.int16, device="cuda")
    y, ms = A.empty_paired((64, 1 << 14), torch.int64, x, candidates=3)
    assert y.shape == (64, 1 << 14) and len(ms) == 3 and all(t > 0 for t in ms)
    assert A.diag_mix_ms(x, y) > 0 and A.diag_mix_ms(y, x) > 0                    # read-dominant and write-dominant
    calls = []
    z, ms = A.shop_output(lambda t: calls.append(t.data_ptr()) or t.zero_(), (64, 1 << 14), torch.int32, x.device, candidates=4, reps=2)
    assert z.shape == (64, 1 << 14) and len(set(calls)) == 4 and len(calls) == 4 * 2 * 3 and z.data_ptr() in calls
