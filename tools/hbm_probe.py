#!/usr/bin/env python3
"""Achievable HBM bandwidth on this part with plain torch kernels (calibration for the roofline fractions)."""
import torch

def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

for gb in (4, 16):
    x = torch.empty(gb * (1 << 30) // 4, dtype=torch.int32, device="cuda").random_()
    y = torch.empty_like(x)
    dt = t(lambda: y.copy_(x))
    print("copy  %2d GiB: %.2f TB/s (read + write)" % (gb, 2 * x.numel() * 4 / dt / 1e12))
    dt = t(lambda: y.fill_(7))
    print("fill  %2d GiB: %.2f TB/s (write only)" % (gb, x.numel() * 4 / dt / 1e12))
    dt = t(lambda: torch.sum(x))
    print("sum   %2d GiB: %.2f TB/s (read only)" % (gb, x.numel() * 4 / dt / 1e12))
    z = torch.empty(x.numel() // 4, dtype=torch.int32, device="cuda")
    xv = x.view(-1, 4)
    dt = t(lambda: torch.sum(xv, dim=1, out=z) if False else z.copy_(xv[:, 0]))
    del x, y, z, xv
