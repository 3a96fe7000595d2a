#!/usr/bin/env python3
"""Per-basic-block instruction mix (blocks that hold MFMAs) of one kernel in a hipcc -S listing.
usage: isa_blocks_mfma.py listing.s <substring of the mangled kernel name>"""
import re
import sys
from collections import Counter
txt = open(sys.argv[1]).read()
m = re.search(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):', txt, re.M)
body = txt[m.end():txt.index('s_endpgm', m.end())]
blocks = re.split(r'\n(\.LBB\d+_\d+):', body)
for i in range(1, len(blocks), 2):
    b = blocks[i + 1]
    n = len(re.findall(r'v_mfma', b))
    if n:
        ins = [l.split()[0] for l in b.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        c = Counter(ins)
        valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
        print(blocks[i], 'mfma', n, 'total', len(ins), 'valu', valu, 'lds', sum(v for k, v in c.items() if k.startswith('ds_')),
              'salu', sum(v for k, v in c.items() if k.startswith('s_') and not k.startswith('s_waitcnt')), 'scratch', sum(v for k, v in c.items() if k.startswith('scratch_')))
        print('   ', sorted(((k, v) for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma')), key=lambda x: -x[1])[:14])
tail = txt[m.end():m.end() + 4000000]
for key in ('NumVgprs', 'Occupancy', 'ScratchSize'):
    mm = re.search(r';\s*' + key + r':\s*(\d+)', tail)
    print(key, mm.group(1))
