#!/usr/bin/env python3
"""Summarise the rocprofv3 databases written by tools/prof.sh into a small text report (for profiles/)."""
import collections
import glob
import os
import sqlite3
import sys


def main(d, out=None):
    lines = []
    tr = os.path.join(d, "trace_results.db")
    if os.path.exists(tr):
        cur = sqlite3.connect(tr).cursor()
        lines.append("== rocprofv3 --kernel-trace --stats (durations in us) ==")
        lines.append("%-90s %6s %12s %10s %6s" % ("kernel", "calls", "total", "avg", "%"))
        for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            lines.append("%-90s %6d %12.1f %10.2f %6.2f" % (name[:90], calls, tot, avg, pct))
    for f in sorted(glob.glob(os.path.join(d, "pmc*_results.db"))):
        cur = sqlite3.connect(f).cursor()
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
            agg[k][c].append(v)
        lines.append("")
        lines.append("== %s: rocprofv3 --pmc (mean per dispatch) ==" % os.path.basename(f))
        for k, cs in agg.items():
            if "fill_stimulus" in k or "rocclr" in k:
                continue
            lines.append(k[:110])
            for c, vals in sorted(cs.items()):
                lines.append("    %-28s n=%-3d %.6g" % (c, len(vals), sum(vals) / len(vals)))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
