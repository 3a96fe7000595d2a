"""Loader for tests/golden/ref_hdr/*.json: raw-integer input / output vectors produced by the REFERENCE's own headers
(tools/gen_golden, build container only).  Shared by the CPU test (oracle vs vectors) and the GPU test (HIP engine vs
vectors)."""
import glob
import json
import os

import numpy as np

HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_hdr")
# ac_mv_avg: the reference's run() / mvAvgCore() loops compiled over this repo's OWN restatement of ac_window_1d_flag (the class
# lives in the absent hlslibs/ac_types): these vectors pin the MAC loop, not the window's boundary rules -- kept apart and named so.
HDR_UNPINNED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_hdr_window_unpinned")


def load(kind):
    """All cases whose "class" is one of `kind` (tuple), as (id, dict) pairs in file order."""
    out = []
    for f in sorted(glob.glob(os.path.join(HDR, "*.json"))) + sorted(glob.glob(os.path.join(HDR_UNPINNED, "*.json"))):
        for c in json.load(open(f))["cases"]:
            if c["class"] in kind:
                out.append(c)
    assert out, "no golden cases for %s under %s" % (kind, HDR)
    return out


def ids(cases):
    return [c["name"] for c in cases]


def arr(c, key):
    return np.array(c[key], dtype=np.int64)


def fmt_of(cls, a):
    """[W, I, S, Q, O] -> Fmt of the oracle or of the engine (both take the same positional fields)."""
    return cls(a[0], a[1], bool(a[2]), a[3], a[4])


def segments(c):
    """(coefficient array, start, stop) runs of a FIR case: one per run() call, the coefficient set switching at reload_at."""
    ra = c.get("reload_at", -1)
    n = len(c["x"])
    cuts = [0]
    for k in c["calls"]:
        cuts.append(cuts[-1] + k)
    assert cuts[-1] == n
    if ra >= 0 and ra not in cuts:
        cuts = sorted(set(cuts + [ra]))
    segs = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        segs.append((arr(c, "coeffs2") if (ra >= 0 and a >= ra) else arr(c, "coeffs"), a, b))
    return segs


def reg_share_taps(c):
    """coeffs[N_TAPS] as ac_fir_reg_share receives it -> tap order (ac_fir_reg_share.h:136-260 addressing)."""
    n, ft = c["n_taps"], c["ftype"]
    count = n if ft == "SHIFT_REG" else (n // 2 if "EVEN" in ft else (n - 1) // 2 + 1)
    raw = arr(c, "coeffs")
    tap = np.zeros(n, dtype=np.int64)
    for t in range(count):
        tap[t] = raw[(t // c["blk_sz"]) * c["mem_word_width"] + c["blk_offset"] + t % c["blk_sz"]]
    return tap
