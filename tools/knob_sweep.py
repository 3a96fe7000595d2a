#!/usr/bin/env python3
"""tools/knob_sweep.py <workload> <passes> VAR=v1|v2|... [VAR2=...] [--libs a.so,b.so] -- same-process, same-allocation A/B of tuning knobs.

Builds ONE bench.py workload (its buffers stay where the driver put them: the streaming rows move 3 - 8 % with the placement of the
input / output pair, profiles/r3_placement_modes.txt), then times every combination of the listed environment knobs round-robin,
`passes` times, 20 steps each behind 3 warm-up steps.  ACDSP_TUNE_LIVE=1 makes the library re-read its knobs at every launch.
Prints per combination: whole-step ms (events on the current stream) of every pass, the minimum and the median, and the roofline
fraction of the median.  A value of `-` means "variable unset".

    python tools/knob_sweep.py cic_dec 3 'ACDSP_GEN_RING=0|4,1,0|4,1,1|4,2,1' 'ACDSP_XCD_MAP=0|1'
"""
import itertools
import os
import sys

os.environ["ACDSP_TUNE_LIVE"] = "1"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    workload, passes = sys.argv[1], int(sys.argv[2])
    knobs = []
    for a in sys.argv[3:]:
        k, v = a.split("=", 1)
        knobs.append((k, v.split("|")))

    class Args:
        channels = 0
        samples = 0
        pad = 0
        stim_bits = 0
    w = bench.build_workload(workload, Args, 1, 0, 0)
    step = w["step"]
    bench.settle_clocks(step, 0.3)
    combos = list(itertools.product(*[vs for _, vs in knobs]))
    res = {c: [] for c in combos}
    for _ in range(passes):
        for c in combos:
            for (k, _), v in zip(knobs, c):
                if v == "-":
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                step()
            e1.record()
            torch.cuda.synchronize()
            res[c].append(e0.elapsed_time(e1) / 20)
    gb = w["samples_per_step"] * w["bytes_per_sample"] / 1e9
    print("# %s: %s  (%.2f GB algorithmic per step; path %s)" % (workload, w["name"], gb, w["path"]))
    for c in combos:
        ts = sorted(res[c])
        med = ts[len(ts) // 2]
        print("%-60s %s  min %.4f  med %.4f ms  %.3f of 8 TB/s" % (
            " ".join("%s=%s" % (k, v) for (k, _), v in zip(knobs, c)), " ".join("%.4f" % t for t in res[c]), ts[0], med, gb / med / 8.0))


if __name__ == "__main__":
    main()
