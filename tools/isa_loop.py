#!/usr/bin/env python3
"""Instruction mix of the hottest basic block (most MFMAs) of one kernel in a hipcc -S listing.

usage: isa_loop.py listing.s <substring of the mangled kernel name>
"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
m = re.search(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):', txt, re.M)
if not m:
    sys.exit("kernel not found")
body = txt[m.end():txt.index('s_endpgm', m.end())]
blocks = re.split(r'\n(\.LBB\d+_\d+):', body)
best = None
for i in range(1, len(blocks), 2):
    n = len(re.findall(r'v_mfma', blocks[i + 1]))
    if best is None or n > best[0]:
        best = (n, blocks[i], blocks[i + 1])
n, lab, b = best
ins = [l.split()[0] for l in b.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
c = Counter(ins)
cat = Counter()
for k, v in c.items():
    if k.startswith('v_mfma'):
        cat['mfma'] += v
    elif k.startswith('v_'):
        cat['valu'] += v
    elif k.startswith('ds_'):
        cat['lds'] += v
    elif k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        cat['vmem'] += v
    elif k.startswith('s_waitcnt'):
        cat['wait'] += v
    elif k.startswith('s_'):
        cat['salu'] += v
    else:
        cat[k] += v
print(m.group(1))
print(lab, len(ins), dict(cat))
print(sorted(c.items(), key=lambda x: -x[1])[:40])
print([l.strip() for l in b.split('\n') if 's_waitcnt' in l])
tail = txt[m.end():m.end() + 4000000]
for key in ('.vgpr_count', '.sgpr_count', 'ScratchSize', 'Occupancy', 'NumVgprs', 'NumAgprs'):
    mm = re.search(r';\s*' + key + r':\s*(\d+)', tail)
    if mm:
        print(key, mm.group(1))
