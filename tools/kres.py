#!/usr/bin/env python3
"""tools/kres.py <hipcc -Rpass-analysis=kernel-resource-usage stderr> [filter] -- one line per kernel: VGPR / AGPR / scratch bytes /
spilled VGPRs / LDS bytes / waves per SIMD.  Kernels with scratch are marked '!'."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].split()[0].strip(" ]")

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void acdsp::", "")
    if flt and flt not in dem:
        continue
    scratch = g(r"ScratchSize \[bytes/lane\]")
    print("%s %-60s vgpr %3d agpr %3d scratch %4d spill %3d lds %6d occ %d" % ("!" if scratch > 0 else " ", dem, g("VGPRs"), g("AGPRs"), scratch,
                                                                           g("VGPRs Spill"), g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")))
