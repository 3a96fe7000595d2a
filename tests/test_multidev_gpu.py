"""Two PHYSICAL devices in one process (one host thread + stream per device, channel slices, no collective).

Every channel is an independent filter object with private state (reference include/ac_dsp/ac_fir_const_coeffs.h:124-127,
ac_cic_full_core.h:71-74,219), so a node's GPUs each take a contiguous slice of the channels and nothing is exchanged.  These tests
drive two handles on two devices from two host threads and check each slice bit for bit against the oracle; they skip on a
one-GPU box (the driver's 8-GPU node runs them).  `bench.py --gpus N` self-launching is covered by test_bench_self_launch below,
which runs on any GPU box (ACDSP_BENCH_ONE_GPU=1 places every rank on device 0)."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

import ac_dsp_amd as A
from helpers import ofmt
from oracle import OracleCic, OracleFir, stimulus

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIN, FC, FA, FO = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
SEED = 0xACD5


def two_devices():
    if A.device_count() < 2:
        pytest.skip("needs two MI355X devices")


def coeffs255():
    import bench
    return bench.windowed_sinc_raw(255, 0.1, FC.F)


def run_threads(fn, n):
    err, th = [], []

    def wrap(i):
        try:
            fn(i)
        except BaseException as e:   # noqa: BLE001 -- reported below, on the main thread
            err.append((i, e))
    for i in range(n):
        th.append(threading.Thread(target=wrap, args=(i,)))
        th[-1].start()
    for t in th:
        t.join()
    assert not err, err


def test_two_devices_two_threads_fir_slices_bit_exact():
    """Config-2 channel slices (255 taps, <16,2> -> <16,2,RND,SAT>) on device 0 and device 1 at the same time, two calls each
    (state carry), every slice against the oracle run over the whole record."""
    two_devices()
    import bench
    n_ch, n = 96, 40960
    c = coeffs255()
    got = [None, None]

    def work(dev):
        lo, hi = bench.shard(n_ch, 2, dev)
        torch.cuda.set_device(dev)
        x = torch.empty((hi - lo, n), dtype=torch.int16, device="cuda:%d" % dev)
        A.fill_stimulus(x, SEED, 16, ch0=lo)
        fir = A.Fir(255, "SHIFT_REG", FIN, FC, FA, FO, n_channels=hi - lo, kind="load", device=dev)
        fir.set_coeffs(c)
        assert fir.path == "mfma_i8"
        k = 24576
        y = torch.cat([fir.run(x[:, :k]), fir.run(x[:, k:])], dim=1)
        got[dev] = y.cpu().numpy().astype(np.int64)

    run_threads(work, 2)
    xo = stimulus(SEED, n_ch, n, 16)
    yo = OracleFir(255, "SHIFT_REG", ofmt(FIN), ofmt(FC), ofmt(FA), ofmt(FO), n_ch=n_ch).run(c, xo)
    assert np.array_equal(np.concatenate(got, axis=0), yo)


def test_state_moves_from_device_0_to_device_1():
    """acdsp_fir_state_get on device 0 -> acdsp_fir_state_set on device 1: the stream continues there bit-exactly."""
    two_devices()
    n_ch, n = 8, 20480
    c = coeffs255()
    xo = stimulus(SEED, n_ch, n, 16)
    yo = OracleFir(255, "SHIFT_REG", ofmt(FIN), ofmt(FC), ofmt(FA), ofmt(FO), n_ch=n_ch).run(c, xo)
    out = []
    blob = None
    for dev, (a, b) in enumerate(((0, 9000), (9000, n))):
        torch.cuda.set_device(dev)
        fir = A.Fir(255, "SHIFT_REG", FIN, FC, FA, FO, n_channels=n_ch, kind="load", device=dev)
        fir.set_coeffs(c)
        if blob is not None:
            fir.set_state(blob)
        x = torch.from_numpy(xo[:, a:b].astype(np.int16)).to("cuda:%d" % dev)
        out.append(fir.run(x).cpu().numpy().astype(np.int64))
        blob = fir.state()
    assert np.array_equal(np.concatenate(out, axis=1), yo)


def test_two_devices_cic_slices_bit_exact():
    two_devices()
    import bench
    n_ch, n = 40, 32768
    fin, fo = A.Fmt(32, 16), A.Fmt(47, 31)
    got = [None, None]

    def work(dev):
        lo, hi = bench.shard(n_ch, 2, dev)
        torch.cuda.set_device(dev)
        x = torch.empty((hi - lo, n), dtype=torch.int32, device="cuda:%d" % dev)
        A.fill_stimulus(x, SEED, 32, ch0=lo)
        got[dev] = A.Cic(False, 8, 1, 5, fin, fo, n_channels=hi - lo, device=dev).run(x).cpu().numpy().astype(np.int64)

    run_threads(work, 2)
    yo = OracleCic(0, 8, 1, 5, ofmt(fin), ofmt(fo), n_ch=n_ch).run(stimulus(SEED, n_ch, n, 32))
    assert np.array_equal(np.concatenate(got, axis=0), yo)


def test_bench_self_launch():
    """`python bench.py --gpus 2` with no launcher spawns its two ranks itself and prints ONE JSON line with n_gpus 2 (the driver's
    plain `python3 bench.py --gpus N` used to fall back to one GPU silently).  On a one-GPU box both ranks share device 0."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if A.device_count() < 2:
        env["ACDSP_BENCH_ONE_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--channels", "64",
                        "--samples", "65536", "--settle", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 64 * 65536 * 3 / (d["ms_per_step"] * 3e-3) / 1e6) < 1e-6 * d["value"]


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def _one_json_line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), "stdout must hold exactly ONE JSON line:\n" + p.stdout[-2000:]
    return json.loads(lines[0])


def _env8():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if A.device_count() < 8:
        env["ACDSP_BENCH_ONE_GPU"] = "1"      # every rank / shard on device 0: the 8-rank code path on a one-GPU box
    return env


def test_bench_eight_ranks_the_drivers_launch_line():
    """The driver's own 8-GPU command -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 8 ...` -- at reduced channels: ONE JSON line on stdout (no gloo chatter, no `secondary` rows), n_gpus 8,
    value = the SUM of the ranks' samples over the MAX of their times."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ch, n, steps = 32, 65536, 3
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", str(steps), "--warmup", "1", "--channels", str(ch), "--samples", str(n),
                        "--settle", "0", "--no-cpu-baseline"], env=_env8(), capture_output=True, text=True, timeout=900)
    d = _one_json_line(p)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["steps"] == steps and "secondary" not in d
    assert abs(d["value"] - 8 * ch * n * steps / (d["ms_per_step"] * steps * 1e-3) / 1e6) < 1e-6 * d["value"]
    assert d["config"]["workload"] and d["metric"]


def test_bench_eight_shards_in_one_process():
    """`bench.py --gpus 8 --inproc`: eight channel slices behind one acdsp_node_* handle (one host thread + stream per shard), same contract."""
    ch, n, steps = 32, 65536, 3
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--inproc", "--steps", str(steps), "--warmup", "1", "--channels", str(ch),
                        "--samples", str(n), "--settle", "0", "--no-cpu-baseline"], env=_env8(), capture_output=True, text=True, timeout=900)
    d = _one_json_line(p)
    assert d["n_gpus"] == 8 and d["steps"] == steps and "secondary" not in d
    assert abs(d["value"] - 8 * ch * n * steps / (d["ms_per_step"] * steps * 1e-3) / 1e6) < 1e-6 * d["value"]
