#!/usr/bin/env python3
"""tools/profiles_index.py -- regenerates profiles/README.md: one line per evidence file (bench lines: workload, ms per step, roofline
fraction; text files: their first comment line), newest round first."""
import glob
import json
import os
import re

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")


def describe(path):
    name = os.path.basename(path)
    if name.endswith(".json"):
        try:
            d = json.loads(open(path).read().strip().splitlines()[-1])
            return "bench.py line: %s, %.4g ms/step, roofline %.3f" % (d["config"]["workload"][:70], d["ms_per_step"], d.get("roofline", {}).get("frac", 0))
        except Exception:  # noqa: BLE001
            return "bench.py output"
    if name.endswith("_rocprof.txt"):
        w = re.sub(r"^r\d+_|_rocprof.txt$", "", name)
        return "rocprofv3 --kernel-trace --stats + four --pmc passes of `bench.py --workload %s` (tools/profile_all.sh)" % w
    for line in open(path, errors="replace"):
        line = line.strip()
        if line:
            line = re.sub(r"^#\s*", "", line)
            line = re.sub(r"^profiles/\S+\s+--\s+", "", line)
            return line[:170]
    return ""


def main():
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "r*_*")) if not f.endswith("README.md"))
    rounds = sorted({os.path.basename(f).split("_")[0] for f in files}, key=lambda r: -int(r[1:]))
    out = ["# profiles/ -- evidence index (regenerate with `python tools/profiles_index.py`)", "",
           "Every number DESIGN.md / BASELINE.md quote comes from one of these files; `gpurun_out/` is scratch.  Newest round first.", ""]
    for r in rounds:
        out += ["## round %s" % r[1:], "", "| file | what it holds |", "|---|---|"]
        for f in files:
            if os.path.basename(f).startswith(r + "_"):
                out.append("| `%s` | %s |" % (os.path.basename(f), describe(f).replace("|", "/")))
        out.append("")
    open(os.path.join(ROOT, "README.md"), "w").write("\n".join(out))


if __name__ == "__main__":
    main()
