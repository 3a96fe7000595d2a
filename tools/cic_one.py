#!/usr/bin/env python3
"""tools/cic_one.py W I R M N [iters] -- one ac_cic_dec_full parameter set, 4096 ch x 2^20 samples, a few launches: the command rocprofv3 wraps
when a shape outside bench.py's rows needs a kernel trace / PMC pass (tools/cic_sweep.py prints the timing table)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ac_dsp_amd as A

W, I, R, M, Ns = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda", 0)
NCH, N = 4096, 1 << 20
fin = A.Fmt(W, I)
it = A.Cic(False, R, M, Ns, fin, fin, n_channels=1, device=0).int_type
fo = A.Fmt(it.W, it.I)
eng = A.Cic(False, R, M, Ns, fin, fo, n_channels=NCH, device=0)
x = torch.empty((NCH, N), dtype=A.torch_dtype_for(fin), device=dev)
A.fill_stimulus(x, 0xACD5, W if W <= 32 else 32, ch0=0)
y = torch.empty((NCH, (N // R + 8 + 7) // 8 * 8), dtype=A.torch_dtype_for(fo), device=dev)
for _ in range(iters):
    eng.run(x, y)
torch.cuda.synchronize()
print("path", eng.path, "kernel ms (avg, min)", eng.kernel_stats(iters))
