"""GPU parity tests, integrate-and-dump: acdsp_intgdump_* vs the oracle restatement of reference
include/ac_dsp/ac_intg_dump.h:93-147 (the reference ships no test or vector for this class)."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from oracle import OracleIntgDump
from helpers import ofmt

pytestmark = pytest.mark.gpu


def rand_raw(rng, fmt, shape):
    lo = -(1 << (fmt.W - 1)) if fmt.S else 0
    hi = (1 << (fmt.W - 1)) - 1 if fmt.S else (1 << fmt.W) - 1
    return rng.integers(lo, hi + 1, size=shape, dtype=np.int64)


def check(ns, chn, fin, fa, fo, calls, n_obj=3, seed=0):
    rng = np.random.default_rng(seed)
    eng = A.IntgDump(ns, chn, fin, fa, fo, n_objects=n_obj)
    orc = OracleIntgDump(ns, chn, ofmt(fin), ofmt(fa), ofmt(fo), n_obj=n_obj)
    for n_sample in calls:
        ni, no = eng.counts(n_sample)
        x = rand_raw(rng, fin, (n_obj, max(ni, 1)))
        y = eng.run(torch.from_numpy(x).to(A.torch_dtype_for(fin)).cuda(), n_sample).cpu().numpy().astype(np.int64)
        yo = orc.run(x, n_sample)
        assert y.shape == yo.shape == (n_obj, no)
        assert np.array_equal(y, yo)


def test_usage_example_types_and_carry_across_blocks_and_calls():
    # reference usage example: <32,16> in, <64,32> out (ac_intg_dump.h:46-50); blocks that dump, blocks whose n_sample is 0
    # or larger than NS (NS rounds, no output, the sums carry into the next block and across run() calls)
    check(16, 4, A.Fmt(32, 16), A.Fmt(64, 32), A.Fmt(64, 32), [[3, 16, 1, 20, 5], [0, 0, 7], [9], [1, 1, 1, 1, 16, 16, 2]], seed=1)


@pytest.mark.parametrize("q,o", [("TRN", "WRAP"), ("RND", "SAT"), ("RND_CONV", "SAT_SYM"), ("TRN_ZERO", "SAT_ZERO")])
def test_every_add_is_an_acc_type_assignment(q, o):
    # narrow accumulator with fewer fraction bits than the input: each `temp[i] + data_in` is quantised and may saturate
    check(64, 3, A.Fmt(16, 4), A.Fmt(18, 8, True, q, o), A.Fmt(10, 7, True, q, o), [[64, 33, 1, 70, 12], [5, 64]], seed=2)


def test_many_objects_long_blocks_unsigned():
    rng = np.random.default_rng(3)
    n_sample = rng.integers(1, 257, size=200)
    check(256, 2, A.Fmt(12, 12, False), A.Fmt(24, 24, False), A.Fmt(24, 24, False), [n_sample], n_obj=64, seed=3)


def test_rejects_short_buffers():
    eng = A.IntgDump(8, 2, A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16))
    x = torch.zeros((1, 8), dtype=torch.int16, device="cuda")
    with pytest.raises(AssertionError):
        eng.run(x, [5])                      # needs 10 samples


# ---- streaming kernel: every block dumps the same number of rounds, CHN divides a 16-byte load ----

@pytest.mark.parametrize("ns,chn,rounds,n_blk,fin,fa,fo", [
    (64, 4, 64, 96, A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16)),                       # bench shape: 32 lanes per block
    (64, 4, 64, 96, A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(12, 6, True, "RND", "SAT")),    # narrowing, saturating OUT_TYPE
    (64, 8, 2, 128, A.Fmt(16, 8, False), A.Fmt(20, 12, False), A.Fmt(20, 12, False)),  # unsigned, 2 lanes per block
    (8, 1, 8, 256, A.Fmt(13, 3), A.Fmt(30, 20), A.Fmt(30, 20)),                        # one lane per block, CHN = 1, F_acc > F_in
    (1024, 2, 1024, 6, A.Fmt(16, 2), A.Fmt(27, 13), A.Fmt(27, 13)),                    # blocks of 4 KB: four loads per lane and reduce
    (256, 4, 256, 16, A.Fmt(32, 16), A.Fmt(44, 28), A.Fmt(44, 28)),                    # int32 containers, 64-bit sums
    (64, 2, 64, 32, A.Fmt(24, 8, False), A.Fmt(34, 18, False), A.Fmt(15, 9, False, "TRN", "WRAP")),
    (48, 4, 48, 64, A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16)),                       # 24 lanes per block: not a power of two -> tiled kernel
    (1000, 4, 1000, 37, A.Fmt(16, 8), A.Fmt(40, 24), A.Fmt(40, 24)),                   # 500 lane-loads per block: ragged blocks, masked eighth load
    (1000, 1, 1000, 41, A.Fmt(16, 8, False), A.Fmt(30, 14, False), A.Fmt(30, 14, False)),   # 125 lane-loads, unsigned
    (100, 4, 100, 75, A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(16, 8, True, "RND", "SAT")),  # 50 lane-loads: one masked load per block
    (36, 8, 36, 130, A.Fmt(12, 4), A.Fmt(24, 12), A.Fmt(24, 12)),                      # 36 lane-loads, eight channels
    (330, 2, 330, 51, A.Fmt(32, 16), A.Fmt(48, 32), A.Fmt(48, 32)),                    # int32 containers: 165 lane-loads
    (1000, 2, 1000, 9, A.Fmt(16, 2), A.Fmt(27, 13, True, "TRN", "WRAP"), A.Fmt(20, 6, True, "RND", "SAT")),   # 250 lane-loads, few blocks
    (1024, 4, 1024, 8, A.Fmt(32, 16), A.Fmt(64, 32), A.Fmt(64, 32)),                   # the header's usage example (ac_intg_dump.h:47-51): 64-bit ACC = OUT, a bit copy
    (64, 2, 64, 32, A.Fmt(32, 16, False), A.Fmt(63, 31, False), A.Fmt(63, 31, False)), # ... unsigned, 63 bits (the C ABI's widest unsigned format): a bit copy
    (64, 4, 64, 32, A.Fmt(32, 16), A.Fmt(64, 32), A.Fmt(32, 16)),                      # 64-bit ACC, truncating right shift into a narrower OUT
    (64, 4, 64, 32, A.Fmt(32, 16), A.Fmt(64, 40), A.Fmt(64, 32)),                      # ... left shift under AC_WRAP
    (64, 4, 64, 32, A.Fmt(32, 16), A.Fmt(64, 32), A.Fmt(40, 20, True, "RND", "SAT")),  # rounding add on 64 bits: stays on the tiled kernel
    (64, 4, 64, 32, A.Fmt(32, 16), A.Fmt(63, 31), A.Fmt(48, 31, True, "TRN", "SAT")),  # 63-bit ACC, clamp without a shift
    (64, 4, 64, 96, A.Fmt(12, 4), A.Fmt(24, 12, True, "TRN", "SAT"), A.Fmt(24, 12)),   # saturating ACC that 64 x 2^11 x 2^4 cannot reach: a wrapping one
    (64, 4, 64, 96, A.Fmt(12, 4), A.Fmt(22, 10, True, "TRN", "SAT_SYM"), A.Fmt(22, 10)),   # ... exactly one LSB inside (2^21 - 1 against 2^21): still saturating
    (64, 4, 64, 96, A.Fmt(16, 8), A.Fmt(20, 12, True, "TRN", "SAT"), A.Fmt(20, 12)),   # ... that random full-scale samples do drive into the rails
    (64, 2, 64, 96, A.Fmt(14, 6, False), A.Fmt(20, 12, False, "TRN", "SAT"), A.Fmt(20, 12, False)),   # unsigned: 64 x (2^14 - 1) < 2^20
])
def test_streaming_kernel_shapes(ns, chn, rounds, n_blk, fin, fa, fo):
    check(ns, chn, fin, fa, fo, [[rounds] * n_blk, [rounds] * (n_blk // 2)], n_obj=5, seed=ns + chn)


# ---- batched streaming kernel (int16 rows, blocks of 128 B .. 1 KB): reduce-scatter over the lanes of a block ----

@pytest.mark.parametrize("chn", [1, 2, 4])
@pytest.mark.parametrize("gs", [8, 16, 32, 64])
def test_batched_kernel_every_group_size(chn, gs):
    rounds = 8 * gs // chn                      # gs lanes x 8 int16 elements per block
    per_load = 64 // gs                         # blocks per 1 KB wave-load
    # 21 and 11 wave-loads per object: batches of 8 with ragged tails, three waves in the first call
    signed = (gs + chn) % 3 != 0
    fin = A.Fmt(16, 8, signed)
    fa = A.Fmt(33, 20, signed)
    fo = [A.Fmt(33, 20, signed), A.Fmt(12, 6, signed, "RND", "SAT"), A.Fmt(31, 19, signed, "TRN", "WRAP")][(gs // 8 + chn) % 3]
    check(rounds, chn, fin, fa, fo, [[rounds] * (21 * per_load), [rounds] * (11 * per_load)], n_obj=3, seed=100 * gs + chn)


def test_batched_kernel_many_waves_full_scale_inputs():
    # every sample at the extreme of <16,8>: the int32 lane sums are at their bound (rounds < 2^15)
    ns, chn, n_obj = 64, 4, 4
    eng = A.IntgDump(ns, chn, A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16), n_objects=n_obj)
    orc = OracleIntgDump(ns, chn, ofmt(A.Fmt(16, 8)), ofmt(A.Fmt(32, 16)), ofmt(A.Fmt(32, 16)), n_obj=n_obj)
    n_sample = [ns] * 2048                      # 1024 wave-loads per object
    x = np.full((n_obj, ns * chn * len(n_sample)), -32768, dtype=np.int64)
    x[1] = 32767
    x[2, ::3] = 32767
    y = eng.run(torch.from_numpy(x).to(torch.int16).cuda(), n_sample).cpu().numpy().astype(np.int64)
    assert np.array_equal(y, orc.run(x, n_sample))


def test_block_table_is_cached_across_calls_and_streams():
    # the same n_sample[] call after call (table reused, run() stays asynchronous), then other tables, other streams, and a
    # growing block count (device arrays reallocated): every call against the oracle, which rebuilds everything every time
    ns, chn, n_obj = 64, 4, 3
    fin, fa, fo = A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16)
    rng = np.random.default_rng(11)
    eng = A.IntgDump(ns, chn, fin, fa, fo, n_objects=n_obj)
    orc = OracleIntgDump(ns, chn, ofmt(fin), ofmt(fa), ofmt(fo), n_obj=n_obj)
    uniform, ragged = [ns] * 32, [3, ns, 1, 70, 12, ns, ns, 5]
    side = torch.cuda.Stream()
    plan = [(uniform, None), (uniform, None), (uniform, side), (ragged, side), (ragged, None), (uniform, None),
            ([ns] * 96, None), ([ns] * 96, side), (uniform, None)]
    for n_sample, stream in plan:
        ni, no = eng.counts(n_sample)
        x = rand_raw(rng, fin, (n_obj, ni))
        xd = torch.from_numpy(x).to(torch.int16).cuda()
        torch.cuda.synchronize()
        if stream is None:
            y = eng.run(xd, n_sample)
        else:
            with torch.cuda.stream(stream):
                y = eng.run(xd, n_sample)
        torch.cuda.synchronize()
        assert np.array_equal(y.cpu().numpy().astype(np.int64), orc.run(x, n_sample))


def test_pending_sums_and_uniform_calls_alternate_on_every_kernel_family():
    """Calls that leave an undumped sum behind (general kernel, temp[] non-zero), calls that dump it and everything after (general kernel again:
    a sum is carried in), then uniform calls (tile / streaming kernels, which leave temp[] alone and rely on it being zero) -- several times
    round, with full-scale samples so that a stale temp[] would show.  acdsp_intgdump_path tells which family ran."""
    ns, chn, n_obj = 64, 4, 5
    fin, fa, fo = A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16)
    rng = np.random.default_rng(21)
    eng = A.IntgDump(ns, chn, fin, fa, fo, n_objects=n_obj)
    orc = OracleIntgDump(ns, chn, ofmt(fin), ofmt(fa), ofmt(fo), n_obj=n_obj)
    uniform, leaves_pending, dumps_pending, ragged_clean = [ns] * 64, [ns, ns, 7, 100], [ns, ns], [5, ns, 9, ns]   # (a block of n_sample > NS runs NS rounds and does not dump)
    seen = []
    for n_sample in [uniform, leaves_pending, dumps_pending, uniform, uniform, leaves_pending, leaves_pending, dumps_pending, ragged_clean, uniform,
                     [ns] * 3, leaves_pending, uniform]:
        ni, no = eng.counts(n_sample)
        x = rand_raw(rng, fin, (n_obj, ni))
        x[0] = 32767
        x[1] = -32768
        y = eng.run(torch.from_numpy(x).to(torch.int16).cuda(), n_sample).cpu().numpy().astype(np.int64)
        seen.append(eng.path)
        assert np.array_equal(y, orc.run(x, n_sample)), (n_sample[:6], eng.path)
    assert seen[0] == "stream" and seen[1] == "exact_order" and seen[2] == "exact_order" and seen[3] == "stream" and seen[9] == "stream", seen
    # (the last call starts with a pending sum: it must NOT take a kernel that ignores temp[])
    assert seen[-1] == "exact_order", seen


# ---- matrix-core kernel (round 6): channel counts that do not divide a 16-byte load; every block the same rounds, a multiple of 16 samples ----

@pytest.mark.parametrize("ns,chn,rounds,n_blk,fin,fa,fo", [
    (64, 7, 64, 40, A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16)),                       # tools/misc_shapes.py: CHN = 7 (0.30 on the tiled kernel)
    (64, 16, 64, 33, A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(16, 8, True, "RND", "SAT")),   # CHN = 16, narrow saturating output
    (64, 3, 64, 50, A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16)),
    (32, 5, 32, 17, A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16)),                       # a partial tile of blocks (17 = 16 + 1)
    (48, 6, 48, 64, A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16)),                       # 288 samples per block: the last segment holds 32
    (16, 9, 16, 100, A.Fmt(12, 4, False), A.Fmt(24, 12, False), A.Fmt(24, 12, False)), # unsigned samples: every plane re-biased
    (80, 11, 80, 31, A.Fmt(16, 1), A.Fmt(40, 20), A.Fmt(40, 20)),                      # F_acc > F_in
    (64, 13, 64, 20, A.Fmt(32, 16), A.Fmt(48, 32), A.Fmt(48, 32)),                     # 32-bit samples: four planes
    (64, 15, 64, 16, A.Fmt(24, 8), A.Fmt(40, 24), A.Fmt(20, 10, True, "RND_CONV", "SAT_SYM")),
    (1024, 7, 1024, 19, A.Fmt(16, 8), A.Fmt(40, 24), A.Fmt(40, 24)),                   # long blocks: 112 segments each
    (64, 12, 64, 48, A.Fmt(16, 8), A.Fmt(31, 15, True, "TRN", "SAT"), A.Fmt(31, 15)),  # a saturating accumulator that cannot saturate (64 x 2^23 < 2^30)
    (16, 10, 16, 64, A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16)),
    (8, 14, 8, 64, A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16)),                        # 112 samples per block
])
def test_matrix_core_kernel_for_channel_counts_that_do_not_divide_a_vector(ns, chn, rounds, n_blk, fin, fa, fo):
    n_obj = 5
    rng = np.random.default_rng(ns * 100 + chn)
    eng = A.IntgDump(ns, chn, fin, fa, fo, n_objects=n_obj)
    orc = OracleIntgDump(ns, chn, ofmt(fin), ofmt(fa), ofmt(fo), n_obj=n_obj)
    for call in range(2):                                   # two calls: nothing is carried, the second call takes the same kernel
        n_sample = [rounds] * n_blk
        ni, no = eng.counts(n_sample)
        x = rand_raw(rng, fin, (n_obj, ni))
        lo, hi = (-(1 << (fin.W - 1)), (1 << (fin.W - 1)) - 1) if fin.S else (0, (1 << fin.W) - 1)
        x[0], x[1] = hi, lo                                  # full-scale rows
        stride = (ni + 63) // 64 * 64                        # 16-byte aligned rows
        xd = torch.zeros((n_obj, stride), dtype=A.torch_dtype_for(fin), device="cuda")
        xd[:, :ni] = torch.from_numpy(x).to(A.torch_dtype_for(fin)).cuda()
        y = eng.run(xd[:, :ni], n_sample).cpu().numpy().astype(np.int64)
        assert eng.path == "mfma", eng.path
        yo = orc.run(x, n_sample)
        assert y.shape == yo.shape == (n_obj, no)
        assert np.array_equal(y, yo), np.argwhere(y != yo)[:5]


def test_matrix_core_kernel_declines_what_it_cannot_take():
    # blocks that are no multiple of 16 samples (7 x 9 = 63), ragged blocks, a carried sum: the tiled / exact-order kernels, same results
    fin, fa, fo = A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16)
    for ns, chn, n_sample in ((9, 7, [9] * 32), (64, 7, [64, 64, 3, 64]), (64, 7, [64, 100, 64, 64])):
        eng = A.IntgDump(ns, chn, fin, fa, fo, n_objects=2)
        orc = OracleIntgDump(ns, chn, ofmt(fin), ofmt(fa), ofmt(fo), n_obj=2)
        ni, no = eng.counts(n_sample)
        x = rand_raw(np.random.default_rng(ns + len(n_sample)), fin, (2, ni))
        y = eng.run(torch.from_numpy(x).to(torch.int16).cuda(), n_sample).cpu().numpy().astype(np.int64)
        assert eng.path != "mfma"
        assert np.array_equal(y, orc.run(x, n_sample))
