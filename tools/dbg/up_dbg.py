import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import ac_dsp_amd as A
from test_polyintr_gpu import check_up
fo = A.Fmt(16, 2, True, "RND", "SAT")
n = int(sys.argv[1]); ifac = int(sys.argv[2])
print("start", n, ifac, flush=True)
eng = check_up(16, ifac, "FOLD_EVEN", fo, n=n, seed=1, pairs=True, expect=None)
torch.cuda.synchronize()
print("done", eng.path, flush=True)
