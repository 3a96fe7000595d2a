#!/usr/bin/env python3
"""Throughput of the CIC interpolator path (no BASELINE config covers it): N5 R8 on <32,16>, 1024 channels."""
import torch
import ac_dsp_amd as A

n_ch, n = 1024, 1 << 18
fin = A.Fmt(32, 16)
probe = A.Cic(True, 8, 1, 5, fin, fin, n_channels=1)
it = probe.int_type
fout = A.Fmt(it.W, it.I)
eng = A.Cic(True, 8, 1, 5, fin, fout, n_channels=n_ch)
x = torch.empty((n_ch, n), dtype=torch.int32, device="cuda")
A.fill_stimulus(x, 1, 32)
y = torch.empty((n_ch, n * 8 + 64), dtype=A.torch_dtype_for(fout), device="cuda")
for _ in range(2):
    eng.reset()
    out = eng.run(x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
K = 5
for _ in range(K):
    out = eng.run(x, y)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
gb = (x.numel() * 4 + out.numel() * out.element_size()) / 1e9
print("cic_intr N5 R8: %d ch x %d in -> %d out/ch, %.2f ms, %.2f GB -> %.2f TB/s (path %s, out type <%d,%d>)" % (n_ch, n, out.shape[1], ms, gb, gb / ms, eng.path, it.W, it.I))
