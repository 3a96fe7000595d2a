import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import ac_dsp_amd as A
from bench import windowed_sinc_raw
fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
nch, n = 1024, 1 << 22
def engine():
    eng = A.PolyDec(16, 8, fin, fc, fa, fo, n_channels=nch)
    hh = np.concatenate([windowed_sinc_raw(127, 0.05, fc.F), [0]])
    eng.set_coeffs(np.array([hh[df + tp * 8] for df in range(8) for tp in range(16)], dtype=np.int64))
    return eng
def timeit(eng, x, y):
    for _ in range(150): eng.run(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): eng.run(x, y)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / 20, 4)
def alloc():
    x = torch.empty((nch, n), dtype=torch.int16, device="cuda"); A.fill_stimulus(x, 7, 16, ch0=0)
    y = torch.empty((nch, n // 8 + 8), dtype=torch.int16, device="cuda")
    return x, y
x, y = alloc(); eng = engine()
print("A: same x,y, new engine each time")
for i in range(5):
    print("  ", timeit(eng, x, y)); eng = engine()
print("B: same engine, new y each time (x kept)")
for i in range(6):
    del y; torch.cuda.empty_cache(); y = torch.empty((nch, n // 8 + 8), dtype=torch.int16, device="cuda")
    print("  ", timeit(eng, x, y), "y %x" % y.data_ptr())
print("C: same engine, new x each time (y kept)")
for i in range(6):
    del x; torch.cuda.empty_cache(); x = torch.empty((nch, n), dtype=torch.int16, device="cuda"); A.fill_stimulus(x, 7, 16, ch0=0)
    print("  ", timeit(eng, x, y), "x %x" % x.data_ptr())
print("D: repeat without realloc")
for i in range(4): print("  ", timeit(eng, x, y))
