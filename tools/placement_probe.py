#!/usr/bin/env python3
"""tools/placement_probe.py <mode> [args] -- why the streaming rows show two speeds (poly_dec 1.69 / 1.83 ms, fused DDC 1.80 / 1.95 ms,
config 3 14.4 / 15.3 ms: every box shows both, a process stays in one).  One process, one engine handle; what varies is where the
driver put the buffers.  Findings: profiles/r3_placement_modes.txt.  (Round 3 grew these as mode_probe.py .. mode_probe5.py.)

  sets <fir255|polydec> [n=6]     n independently allocated (input, output) sets visited round-robin, then the x_i / y_j matrix:
                                  does the speed follow a buffer, the pair, or the process?
  offset                          poly_dec; input and output carved out of ONE allocation, the output at a swept byte offset behind
                                  the input (2 MB .. 256 B steps), then separate output allocations
  far [pool GB=16] [step MB=32]   poly_dec; the same at distances of up to many GB inside one allocation
  outputs <polydec|ddc|cic_dec|mvavg> [k=8]   one input, k separately allocated outputs, the row's time with each (with the same
                                  allocation sequence two builds, ACDSP_LIB=..., see comparable placements: an A/B per placement)
  pmc                             poly_dec; as `outputs`, then the slowest and the fastest pair 8 launches each, slow first -- the LAST 16
                                  dispatches of the kernel in a rocprofv3 --pmc counter CSV
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ac_dsp_amd as A  # noqa: E402
from bench import windowed_sinc_raw  # noqa: E402

DEV = torch.device("cuda", 0)


def workload(name):
    """(engine, run(x, y), NCH, N, input dtype, stimulus bits, output dtype, output row length)"""
    fin, fc, fa, fo = A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2, True, "RND", "SAT")
    if name == "fir255":
        nch, n = 1024, 1 << 20
        eng = A.Fir(255, "SHIFT_REG", fin, fc, fa, fo, n_channels=nch, kind="load", device=0)
        eng.set_coeffs(windowed_sinc_raw(255, 0.1, fc.F))
        return eng, eng.run, nch, n, torch.int16, 16, torch.int16, n
    if name == "polydec":
        nch, n = 1024, 1 << 22
        eng = A.PolyDec(16, 8, fin, fc, fa, fo, n_channels=nch, device=0)
        hh = np.concatenate([windowed_sinc_raw(127, 0.05, fc.F), [0]])
        eng.set_coeffs(np.array([hh[df + tp * 8] for df in range(8) for tp in range(16)], dtype=np.int64))
        return eng, eng.run, nch, n, torch.int16, 16, torch.int16, n // 8 + 8
    if name == "ddc":
        nch, n = 4096, 1 << 20
        dc, da, do = A.Fmt(16, 1), A.Fmt(60, 30), A.Fmt(24, 9, True, "RND", "SAT")
        eng = A.Ddc(16, 1, 5, A.Fmt(16, 1), 127, "SHIFT_REG", dc, da, do, n_channels=nch, kind="const", device=0)
        eng.set_coeffs(windowed_sinc_raw(127, 0.2, dc.F))
        return eng, eng.run, nch, n, torch.int16, 16, torch.int32, n // 16 + 8
    if name == "mvavg":
        nch, n = 1024, 1 << 20
        mc = A.Fmt(16, 2)
        eng = A.MvAvg(1024, 9, "MIRROR", A.Fmt(16, 8), mc, A.Fmt(40, 18), A.Fmt(16, 8, True, "RND", "SAT"), n_objects=nch, device=0)
        eng.set_coeffs(np.round(np.hanning(11)[1:-1] / np.hanning(11).sum() * 2.0 ** mc.F).astype(np.int64))
        return eng, (lambda x, y: eng.run(x, 1024, out=y)), nch, n, torch.int16, 16, torch.int16, n
    nch, n = 4096, 1 << 22
    eng = A.Cic(False, 8, 1, 5, A.Fmt(32, 16), A.Fmt(47, 31), n_channels=nch, device=0)
    return eng, eng.run, nch, n, torch.int32, 32, torch.int64, n // 8 + 8


def timed(run, x, y, reps, warm=2):
    for _ in range(warm):
        run(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(x, y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def mode_sets(argv):
    _, run, nch, n, xdt, bits, ydt, nout = workload(argv[0] if argv else "polydec")
    nsets = int(argv[1]) if len(argv) > 1 else 6
    sets, keep = [], []
    for i in range(nsets):
        x = torch.empty((nch, n), dtype=xdt, device=DEV)
        A.fill_stimulus(x, 0xACD5 + i, bits, ch0=0)
        y = torch.empty((nch, nout), dtype=ydt, device=DEV)
        keep.append(torch.empty((37 + 11 * i) << 20, dtype=torch.uint8, device=DEV))   # odd-sized spacer: moves the next set
        sets.append((x, y))
    for _ in range(300):
        run(*sets[0])
    for rnd in range(4):
        out = [timed(run, x, y, 10, 3) for x, y in sets]
        print("round %d: " % rnd + "  ".join(("set%d %.3f ms (x %#x y %#x)" % (i, t, sets[i][0].data_ptr(), sets[i][1].data_ptr())) if rnd == 0
                                              else "set%d %.3f" % (i, t) for i, t in enumerate(out)))
    print("matrix rows x_i, cols y_j (ms):")
    for i in range(nsets):
        print("  x%d: " % i + " ".join("%.3f" % timed(run, sets[i][0], sets[j][1], 8) for j in range(nsets)))


def carve(pool, nch, n, nout, xdt, ydt, bits):
    xb = nch * n * torch.empty((), dtype=xdt).element_size()
    yb = nch * nout * torch.empty((), dtype=ydt).element_size()
    x = pool[:xb].view(xdt).view(nch, n)
    A.fill_stimulus(x, 0xACD5, bits, ch0=0)
    return x, xb, yb, (lambda d: pool[xb + d: xb + d + yb].view(ydt).view(nch, nout))


def mode_offset(argv):
    _, run, nch, n, xdt, bits, ydt, nout = workload("polydec")
    es = torch.empty((), dtype=xdt).element_size()
    pool = torch.empty(nch * n * es + nch * nout * 2 + (1 << 30), dtype=torch.uint8, device=DEV)
    x, _, _, y_at = carve(pool, nch, n, nout, xdt, ydt, bits)
    for _ in range(200):
        run(x, y_at(0))
    print("pool at %#x" % pool.data_ptr())
    for name, step, cnt in (("2 MB", 2 << 20, 48), ("128 KB", 128 << 10, 32), ("4 KB", 4 << 10, 32), ("256 B", 256, 16)):
        print("step %s: " % name + " ".join("%.3f" % timed(run, x, y_at(k * step), 8) for k in range(cnt)))
    print("repeat D=0: " + " ".join("%.3f" % timed(run, x, y_at(0), 8) for _ in range(6)))
    keep = []
    for i in range(6):
        ysep = torch.empty((nch, nout), dtype=ydt, device=DEV)
        keep.append((ysep, torch.empty((5 + 13 * i) << 20, dtype=torch.uint8, device=DEV)) if i % 2 else None)
        print("separate output allocation %d at %#x: %.3f ms" % (i, ysep.data_ptr(), timed(run, x, ysep, 8)))


def mode_far(argv):
    _, run, nch, n, xdt, bits, ydt, nout = workload("polydec")
    pool_b = (int(argv[0]) if argv else 16) << 30
    step = (int(argv[1]) if len(argv) > 1 else 32) << 20
    pool = torch.empty(pool_b, dtype=torch.uint8, device=DEV)
    x, xb, yb, y_at = carve(pool, nch, n, nout, xdt, ydt, bits)
    for _ in range(200):
        run(x, y_at(0))
    print("pool at %#x, %d GB, step %d MB; one line per GB of distance" % (pool.data_ptr(), pool_b >> 30, step >> 20))
    d, line = 0, []
    while xb + d + yb <= pool_b:
        line.append(timed(run, x, y_at(d), 6, 1))
        d += step
        if d % (1 << 30) == 0:
            print("%3d GB: " % ((d >> 30) - 1) + " ".join("%.2f" % t for t in line))
            line = []
    if line:
        print("  tail: " + " ".join("%.2f" % t for t in line))


def candidates(name, k):
    _, run, nch, n, xdt, bits, ydt, nout = workload(name)
    x = torch.empty((nch, n), dtype=xdt, device=DEV)
    A.fill_stimulus(x, 0xACD5, bits, ch0=0)
    ys, keep = [], []
    for i in range(k):
        ys.append(torch.empty((nch, nout), dtype=ydt, device=DEV))
        keep.append(torch.empty((5 + 13 * i) << 20, dtype=torch.uint8, device=DEV))
    for _ in range(60):
        run(x, ys[0])
    return run, x, ys, [timed(run, x, y, 6) for y in ys], keep


def mode_outputs(argv):
    name = argv[0] if argv else "polydec"
    _, _, _, ts, _ = candidates(name, int(argv[1]) if len(argv) > 1 else 8)
    print("%s candidates: " % name + " ".join("%.3f" % t for t in ts) + "   mean %.3f" % (sum(ts) / len(ts)))


def mode_pmc(argv):
    run, x, ys, ts, _ = candidates("polydec", 8)
    print("candidates: " + " ".join("%.3f" % t for t in ts))
    slow, fast = int(np.argmax(ts)), int(np.argmin(ts))
    print("slow = %d (%#x), fast = %d (%#x), x at %#x" % (slow, ys[slow].data_ptr(), fast, ys[fast].data_ptr(), x.data_ptr()))
    for y in (ys[slow], ys[fast]):
        torch.cuda.synchronize()
        for _ in range(8):
            run(x, y)
        torch.cuda.synchronize()
    print("final: slow %.3f fast %.3f" % (timed(run, x, ys[slow], 6), timed(run, x, ys[fast], 6)))


if __name__ == "__main__":
    modes = {"sets": mode_sets, "offset": mode_offset, "far": mode_far, "outputs": mode_outputs, "pmc": mode_pmc}
    if len(sys.argv) < 2 or sys.argv[1] not in modes:
        raise SystemExit(__doc__)
    modes[sys.argv[1]](sys.argv[2:])
