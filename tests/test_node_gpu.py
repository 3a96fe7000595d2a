"""Node-level sharding as a product entry point (acdsp_node_*): one filter bank cut into contiguous channel slices, one
engine handle + stream + host thread per shard, coefficients replicated, no collective.

The shards of these tests sit on ONE device (the device list repeats device 0), which exercises everything but the physical second
GPU on the one-GPU test box; the last test spreads the shards over every visible device and needs two.  Every slice is checked bit
for bit against the oracle run over the whole, unsharded bank (every channel is an independent filter object with private state:
reference include/ac_dsp/ac_fir_const_coeffs.h:124-127, ac_cic_full_core.h:71-74,219)."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from helpers import ofmt
from oracle import OracleCic, OracleFir, stimulus

pytestmark = pytest.mark.gpu

FIN, FC, FA, FO = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
SEED = 0xACD5


def fill(bits):
    return lambda t, lo: A.fill_stimulus(t, SEED, bits, ch0=lo)


def gather(outs):
    return np.concatenate([o.cpu().numpy().astype(np.int64) for o in outs], axis=0)


@pytest.mark.parametrize("n_shards", [1, 3, 8])
def test_fir_bank_sharded_on_one_device_matches_the_unsharded_oracle(n_shards):
    """BASELINE config 2's filter (255 taps, <16,2> -> <16,2,RND,SAT>) on 50 channels (ragged slices), two calls (state carry)."""
    import bench
    n_ch, n = 50, 6144
    c = bench.windowed_sinc_raw(255, 0.1, FC.F)
    node = A.NodeFir(255, "SHIFT_REG", FIN, FC, FA, FO, n_ch, [0] * n_shards, kind="load")
    assert node.slices == [A.node_shard(n_ch, n_shards, s) for s in range(n_shards)]
    node.set_coeffs(c)
    xs = node.alloc(FIN, 2 * n, fill(16))
    y1 = gather(node.run([x[:, :n] for x in xs]))
    y2 = gather(node.run([x[:, n:] for x in xs]))
    per, mx = node.last_ms()
    assert len(per) == n_shards and mx == max(per) and mx > 0
    want = OracleFir(255, "SHIFT_REG", ofmt(FIN), ofmt(FC), ofmt(FA), ofmt(FO), n_ch=n_ch).run(c, stimulus(SEED, n_ch, 2 * n, 16))
    assert np.array_equal(np.concatenate([y1, y2], axis=1), want)


def test_fir_per_channel_coefficient_sets_are_sliced_and_host_rows_too():
    rng = np.random.default_rng(5)
    n_ch, n_taps, n = 11, 63, 700
    c = rng.integers(-3000, 3000, size=(n_ch, n_taps), dtype=np.int64)
    node = A.NodeFir(n_taps, "C_BUFF", FIN, FC, FA, A.Fmt(40, 12), n_ch, [0, 0, 0, 0], kind="prog", coeffs_per_channel=True)
    node.set_coeffs(c)
    x = stimulus(SEED, n_ch, n, 16)
    y = node.run_host(x.astype(np.int16)).astype(np.int64)
    o = OracleFir(n_taps, "C_BUFF", ofmt(FIN), ofmt(FC), ofmt(FA), ofmt(A.Fmt(40, 12)), n_ch=n_ch)   # a [n_ch][n_taps] array = one set per channel
    assert np.array_equal(y, o.run(c, x))


def test_cic_decimator_bank_sharded_on_one_device():
    """BASELINE config 3's decimator (N5 R8 on <32,16>), ragged call lengths: every shard stays in the same phase."""
    cin, cout = A.Fmt(32, 16), A.Fmt(47, 31)
    n_ch = 21
    node = A.NodeCic(False, 8, 1, 5, cin, cout, n_ch, [0, 0, 0])
    ora = OracleCic(0, 8, 1, 5, ofmt(cin), ofmt(cout), n_ch=n_ch)
    xs = node.alloc(cin, 9000, fill(32))
    xo = stimulus(SEED, n_ch, 9000, 32)
    pos = 0
    for k in (4099, 13, 4888):
        got = gather(node.run([x[:, pos:pos + k] for x in xs]))
        assert np.array_equal(got, ora.run(xo[:, pos:pos + k]))
        pos += k


def test_fused_ddc_bank_sharded_on_one_device():
    """BASELINE config 5's cascade (CIC R16 N5 on <16,1> -> 127-tap FIR on the 36-bit INT_TYPE) through the fused kernel per shard."""
    import bench
    from test_ddc_gpu import oracle_cascade
    cin, fc, fa, fo = A.Fmt(16, 1), A.Fmt(16, 1), A.Fmt(60, 30), A.Fmt(24, 9, True, "RND", "SAT")
    n_ch, n = 10, 16 * 4096
    c = bench.windowed_sinc_raw(127, 0.2, fc.F)
    node = A.NodeDdc(16, 1, 5, cin, 127, "SHIFT_REG", fc, fa, fo, n_ch, [0, 0])
    node.set_coeffs(c)
    xs = node.alloc(cin, n, fill(16))
    got = gather(node.run(xs))
    want = oracle_cascade(16, 1, 5, cin, node.int_type, 127, "SHIFT_REG", fc, fa, fo, c, stimulus(SEED, n_ch, n, 16))
    assert np.array_equal(got, want)


def test_node_argument_errors_are_loud():
    with pytest.raises(A.AcdspError):
        A.NodeFir(31, "SHIFT_REG", FIN, FC, FA, FO, 2, [0, 0, 0])          # fewer channels than shards
    with pytest.raises(A.AcdspError):
        A.NodeFir(31, "SHIFT_REG", FIN, FC, FA, FO, 8, [0, 99])            # no such device
    node = A.NodeFir(31, "SHIFT_REG", FIN, FC, FA, FO, 8, [0, 0])
    xs = node.alloc(FIN, 256, fill(16))
    with pytest.raises(A.AcdspError, match="shard"):                           # run before set_coeffs: the shard's error comes through
        node.run(xs)


def test_shards_on_every_visible_device():
    nd = A.device_count()
    if nd < 2:
        pytest.skip("needs two MI355X devices")
    import bench
    n_ch, n = 64 * nd + 3, 8192
    c = bench.windowed_sinc_raw(255, 0.1, FC.F)
    node = A.NodeFir(255, "SHIFT_REG", FIN, FC, FA, FO, n_ch, list(range(nd)), kind="load")
    node.set_coeffs(c)
    xs = node.alloc(FIN, n, fill(16))
    got = gather(node.run(xs))
    want = OracleFir(255, "SHIFT_REG", ofmt(FIN), ofmt(FC), ofmt(FA), ofmt(FO), n_ch=n_ch).run(c, stimulus(SEED, n_ch, n, 16))
    assert np.array_equal(got, want)
