#!/usr/bin/env python3
"""tools/isa_loops.py file.s name-substring -- per loop (backward branch to a label) of the matching functions: instruction counts by
class (VALU / MFMA / SALU / LDS / VMEM / scratch / waitcnt).  Input: hipcc -S --cuda-device-only output."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    if not (m and pat in m.group(1)):
        i += 1
        continue
    name = m.group(1)
    j = i + 1
    while j < len(lines) and "s_endpgm" not in lines[j]:
        j += 1
    body = lines[i:j + 1]
    labels = {}
    for k, l in enumerate(body):
        mm = re.match(r"^(\.LBB\w+):", l)
        if mm:
            labels[mm.group(1)] = k
    print(name[:110])
    tot_scr = sum(1 for l in body if "scratch_" in l)
    print("  whole function: %d lines, %d scratch ops" % (len(body), tot_scr))
    for k, l in enumerate(body):
        mm = re.search(r"s_cbranch_\w+ (\.LBB\w+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
            seg = [x.strip() for x in body[labels[mm.group(1)]:k + 1] if x.startswith("\t") and not x.strip().startswith((";", "."))]
            c = {"valu": 0, "mfma": 0, "salu": 0, "lds": 0, "vmem": 0, "scratch": 0, "wait": 0}
            for x in seg:
                op = x.split()[0]
                if "mfma" in op: c["mfma"] += 1
                elif op.startswith("scratch_"): c["scratch"] += 1
                elif op.startswith("v_"): c["valu"] += 1
                elif op.startswith("s_waitcnt"): c["wait"] += 1
                elif op.startswith("s_"): c["salu"] += 1
                elif op.startswith("ds_"): c["lds"] += 1
                elif op.startswith(("global_", "buffer_", "flat_")): c["vmem"] += 1
            print("  loop %s: %d instr  %s" % (mm.group(1), len(seg), c))
    i = j
