import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import ac_dsp_amd as A
from oracle import OracleIntgDump
from helpers import ofmt
from test_intgdump_gpu import rand_raw
ns, chn, rounds, n_blk = 64, 2, 64, 32
fin, fa, fo = A.Fmt(24, 8, False), A.Fmt(34, 18, False), A.Fmt(16, 10, False, "TRN", "WRAP")
rng = np.random.default_rng(1)
eng = A.IntgDump(ns, chn, fin, fa, fo, n_objects=2)
orc = OracleIntgDump(ns, chn, ofmt(fin), ofmt(fa), ofmt(fo), n_obj=2)
n_sample = [rounds] * n_blk
ni, no = eng.counts(n_sample)
x = rand_raw(rng, fin, (2, ni))
y = eng.run(torch.from_numpy(x).to(A.torch_dtype_for(fin)).cuda(), n_sample).cpu().numpy().astype(np.int64)
yo = orc.run(x, n_sample)
bad = np.argwhere(y != yo)
print(len(bad), bad[:5], [ (hex(y[tuple(b)]), hex(yo[tuple(b)])) for b in bad[:5]])
