"""tools/clock_ramp.py [workload=fir255] [idle_seconds=0] -- step time against elapsed time from an idle GPU: blocks of 20 steps for
the first 400 steps, then blocks of 200 (HIP events per block).  Evidence for bench.py's `clock_settle` pre-conditioning: the first
block from idle is ~40 % slow (shader clock ramping from ~100 MHz), every later block sits at the steady-state time."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, argparse
import bench
ap = argparse.Namespace(channels=0, samples=0, pad=0, stim_bits=0)
wl = sys.argv[1] if len(sys.argv) > 1 else "fir255"
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
w = bench.build_workload(wl, ap, 1, 0, 0)
torch.cuda.synchronize()
if idle: time.sleep(idle)
step = w["step"]
out = []
t00 = time.perf_counter()
for blk in range(60):
    n = 20 if blk < 20 else 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): step()
    e1.record(); torch.cuda.synchronize()
    out.append((round(time.perf_counter() - t00, 3), n, round(e0.elapsed_time(e1) / n, 4)))
print(wl, "idle", idle)
for o in out: print(o)
