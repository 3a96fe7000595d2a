#!/usr/bin/env python3
"""tools/placement_rank.py <workload> [k=6] -- does the placement probe (acdsp_diag_mix_ms) rank candidate OUTPUT allocations the way the operator
itself does?  One input, k separately allocated outputs: probe time and the row's kernel time with each, their rank correlation, and what
empty_paired() would have picked.  Workloads: cic_dec (config 3), cic_dec_r64, polydec, ddc."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ac_dsp_amd as A  # noqa: E402
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cic_dec"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6


class Args:
    channels = 0
    samples = 0
    pad = 0
    stim_bits = 0


w = bench.build_workload(name, Args, 1, 0, 0)
eng, x = w["eng"], w["x"]
step = w["step"]
step()
torch.cuda.synchronize()
# the workload's own output tensor: find its shape through a run
y0 = eng.run(x) if name.startswith("cic") else None
if y0 is None:
    raise SystemExit("only the cic workloads are wired here")
shape, dt = (y0.shape[0], y0.shape[1] + 8), y0.dtype
del y0
ys = [torch.empty(shape, dtype=dt, device=x.device) for _ in range(k)]


def kernel_ms(y):
    for _ in range(3):
        eng.run(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        eng.run(x, y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 8


cap = int(os.environ.get("PROBE_CAP_GB", "4")) << 30
f = min(1.0, cap / (x.numel() * x.element_size()))
xf = x.reshape(-1)[: int(x.numel() * f)] if x.is_contiguous() else None
for rnd in range(2):
    pr, kr = [], []
    for y in ys:
        yf = y.view(-1)[: int(y.numel() * f)]
        pr.append(min(A.diag_mix_ms(xf, yf), A.diag_mix_ms(xf, yf)))
        kr.append(kernel_ms(y))
    pr, kr = np.array(pr), np.array(kr)
    rp, rk = np.argsort(np.argsort(pr)), np.argsort(np.argsort(kr))
    rho = 1 - 6 * ((rp - rk) ** 2).sum() / (k * (k * k - 1))
    print("round %d  probe ms  %s" % (rnd, " ".join("%.4f" % v for v in pr)))
    print("         kernel ms %s   spearman %.2f   probe's pick %.4f   best %.4f   worst %.4f   mean %.4f" % (
        " ".join("%.4f" % v for v in kr), rho, kr[pr.argmin()], kr.min(), kr.max(), kr.mean()))
