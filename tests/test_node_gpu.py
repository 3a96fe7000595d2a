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
from oracle import OracleCic, OracleFir, OracleIntgDump, OracleMvAvg, OraclePolyDec, OraclePolyIntr, stimulus

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


@pytest.mark.parametrize("n_shards", [1, 3, 8])
def test_f_row_banks_sharded_on_one_device(n_shards):
    """The SURVEY 8(f) classes behind the same node layer (round 5): ac_poly_dec, ac_poly_intr, ac_intg_dump, ac_mv_avg -- rows (channels or
    objects) in contiguous slices, coefficients / control words / block counts replicated, two calls each (state carry where the class has
    state), every slice against the unsharded oracle."""
    import bench
    devs = [0] * n_shards
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
    n_ch = 19
    # ac_poly_dec: 16 taps per branch x DF 8 (the bench row's filter)
    hh = np.concatenate([bench.windowed_sinc_raw(127, 0.05, fc.F), [0]])
    cd = np.array([hh[df + tp * 8] for df in range(8) for tp in range(16)], dtype=np.int64)
    node = A.NodePolyDec(16, 8, fin, fc, fa, fo, n_ch, devs)
    node.set_coeffs(cd)
    ora = OraclePolyDec(16, 8, ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    xs = node.alloc(fin, 8 * 2600, fill(16))
    xo = stimulus(SEED, n_ch, 8 * 2600, 16)
    pos = 0
    for k in (8 * 2048, 8 * 552):
        assert np.array_equal(gather(node.run([x[:, pos:pos + k] for x in xs])), ora.run(cd, xo[:, pos:pos + k]))
        pos += k
    # ac_poly_intr: FOLD_EVEN, 16 taps x IF 8 (the bench row's filter)
    ci = bench.windowed_sinc_raw(127, 0.05, fc.F)[:64]
    node = A.NodePolyIntr(16, 64, 8, "FOLD_EVEN", fin, fc, fa, fo, n_ch, devs)
    node.set_ctrl(ci, [1] * 8, list(range(8)))
    ora = OraclePolyIntr(16, 64, 8, "FOLD_EVEN", ofmt(fin), ofmt(fc), ofmt(fa), ofmt(fo), n_ch=n_ch)
    xs = node.alloc(fin, 16 * 80, fill(16))
    xo = stimulus(SEED, n_ch, 16 * 80, 16)
    pos = 0
    for k in (16 * 70, 16 * 10):
        assert np.array_equal(gather(node.run([x[:, pos:pos + k] for x in xs])), ora.run(ci, [1] * 8, list(range(8)), xo[:, pos:pos + k]))
        pos += k
    # ac_intg_dump: 4 interleaved channels, a block that carries its sums on (n_sample = 0) between dumping ones
    gin, ga, go = A.Fmt(16, 8), A.Fmt(32, 16), A.Fmt(32, 16)
    node = A.NodeIntgDump(64, 4, gin, ga, go, n_ch, devs)
    ora = OracleIntgDump(64, 4, ofmt(gin), ofmt(ga), ofmt(go), n_obj=n_ch)
    ns = np.array([64] * 9 + [0] + [17, 64], dtype=np.int64)
    n_in = int(sum(64 if (v < 1 or v > 64) else v for v in ns)) * 4
    xs = node.alloc(gin, (n_in + 7) // 8 * 8, fill(16))
    xo = stimulus(SEED, n_ch, (n_in + 7) // 8 * 8, 16)
    for _ in range(2):
        assert np.array_equal(gather(node.run(xs, ns)), ora.run(xo[:, :n_in], ns))
    # ac_mv_avg: 9 taps, AC_MIRROR, frames of 256 samples
    mo, mc, ma = A.Fmt(16, 8, True, "RND", "SAT"), A.Fmt(16, 2), A.Fmt(40, 18)
    wts = np.round(np.hanning(11)[1:-1] / np.hanning(11).sum() * 2.0 ** mc.F).astype(np.int64)
    node = A.NodeMvAvg(1024, 9, "MIRROR", gin, mc, ma, mo, n_ch, devs)
    node.set_coeffs(wts)
    ora = OracleMvAvg(9, "MIRROR", ofmt(gin), ofmt(mc), ofmt(ma), ofmt(mo), n_obj=n_ch)
    xs = node.alloc(gin, 256 * 12, fill(16))
    xo = stimulus(SEED, n_ch, 256 * 12, 16)
    assert np.array_equal(gather(node.run(xs, 256)), ora.run(wts, xo, 256))
    per, mx = node.last_ms()
    assert len(per) == n_shards and mx > 0


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
