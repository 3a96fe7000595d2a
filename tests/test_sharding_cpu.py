"""Multi-rank path on CPU (gloo, world_size 2): bench.py's channel-slice sharding.

Channels are independent filter objects, so the N-GPU job is N disjoint channel slices and no data-path
collective.  The test runs the same slicing code under torch.distributed with the CPU oracle standing in
for the per-rank engine (the HIP engine needs a GPU) and checks that the gathered slices equal the
unsharded result and that the timing reduction (MAX over ranks) and the sample count (SUM) behave."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from oracle import OracleFir, Fmt, stimulus
    lo, hi = bench.shard(n_total, world, rank)
    fi, fc, fa, fo = Fmt(16, 2), Fmt(16, 2), Fmt(40, 12), Fmt(16, 2, True, "RND", "SAT")
    c = bench.windowed_sinc_raw(31, 0.1, 14)
    x = stimulus(0xACD5, hi - lo, 256, 16, ch0=lo)          # same generator call bench.py makes per rank
    y = OracleFir(31, "SHIFT_REG", fi, fc, fa, fo, n_ch=hi - lo).run(c, x)
    dist.barrier()
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tot = torch.tensor([float((hi - lo) * 256)], dtype=torch.float64)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, y))
    if rank == 0:
        q.put((float(t.item()), float(tot.item()), gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_channel_slices_cover_and_match_unsharded():
    sys.path.insert(0, ROOT)
    import bench
    from oracle import OracleFir, Fmt, stimulus
    for n_total, world in ((1024, 8), (10, 4), (7, 2), (3, 8)):
        cuts = [bench.shard(n_total, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n_total
        assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
        assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1

    world, n_total = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    tmax, tot, gathered = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 1.5 and tot == n_total * 256
    fi, fc, fa, fo = Fmt(16, 2), Fmt(16, 2), Fmt(40, 12), Fmt(16, 2, True, "RND", "SAT")
    full = OracleFir(31, "SHIFT_REG", fi, fc, fa, fo, n_ch=n_total).run(bench.windowed_sinc_raw(31, 0.1, 14),
                                                                         stimulus(0xACD5, n_total, 256, 16))
    got = np.concatenate([g[2] for g in sorted(gathered, key=lambda g: g[0])], axis=0)
    assert np.array_equal(got, full)


def test_bench_self_launch_gives_every_rank_the_launcher_environment(monkeypatch):
    """`python bench.py --gpus N` with no launcher spawns N ranks itself: every child must see the environment a
    `torch.distributed.run --nproc-per-node N` launch would give it (one rank per GPU, loopback rendezvous)."""
    sys.path.insert(0, ROOT)
    import subprocess

    import bench
    seen = []

    class FakeProc:
        def __init__(self, cmd, env):
            seen.append((cmd, env))

        def wait(self):
            return 0

    monkeypatch.setattr(subprocess, "Popen", lambda cmd, env=None: FakeProc(cmd, env))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    assert bench.self_launch(4) == 0
    assert len(seen) == 4
    ports = {e["MASTER_PORT"] for _, e in seen}
    assert len(ports) == 1 and all(e["MASTER_ADDR"] == "127.0.0.1" and e["WORLD_SIZE"] == "4" for _, e in seen)
    assert [e["RANK"] for _, e in seen] == ["0", "1", "2", "3"] and [e["LOCAL_RANK"] for _, e in seen] == ["0", "1", "2", "3"]
    assert all(cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "7"] for cmd, _ in seen)
