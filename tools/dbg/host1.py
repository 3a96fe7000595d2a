import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import ac_dsp_amd as A
fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
for taps in (63, 255):
    e = A.Fir(taps, "SHIFT_REG", fin, fc, fa, fo, n_channels=1, kind="prog")
    c = np.arange(taps, dtype=np.int64) - taps // 2
    e.set_coeffs(c)
    x = np.array([[123]], dtype=np.int16)
    for _ in range(50): e.run_host(x)
    t0 = time.perf_counter()
    for _ in range(2000): e.run_host(x)
    dt = (time.perf_counter() - t0) / 2000
    t0 = time.perf_counter()
    for _ in range(2000): e.set_coeffs(c); e.run_host(x)
    dt2 = (time.perf_counter() - t0) / 2000
    print("taps %d: run_host(1 sample) %.1f us; set_coeffs + run_host %.1f us" % (taps, dt * 1e6, dt2 * 1e6))
