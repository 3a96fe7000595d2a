import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, argparse
import bench
ap = argparse.Namespace(channels=0, samples=0, pad=0, stim_bits=0)
wl = sys.argv[1] if len(sys.argv) > 1 else "polydec"
junk = []
for it in range(8):
    w = bench.build_workload(wl, ap, 1, 0, 0)
    dt, ka, km, ev = bench.measure(w, 20, 5, lambda: None, 0.2)
    print(wl, it, "x ptr %x" % w["x"].data_ptr(), round(ka, 4), round(ev, 4), flush=True)
    del w
    if it % 2 == 0:
        junk.append(torch.empty((it + 1) * 300_000_000, dtype=torch.uint8, device="cuda"))   # shift the next allocation
    torch.cuda.empty_cache()
