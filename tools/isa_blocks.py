#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing (blocks with >= MIN MFMAs).

usage: isa_blocks.py listing.s <substring of the mangled kernel name> [min_mfma]
"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
name = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m = re.search(r'^(\S*' + re.escape(name) + r'\S*):', txt, re.M)
if not m:
    sys.exit("kernel not found")
end = txt.index('.Lfunc_end', m.end())
body = txt[m.end():end]
tail = txt[end:end + 20000]
info = {}
for k in ('NumVgprs', 'NumAgprs', 'ScratchSize', 'Occupancy'):
    mm = re.search(r';\s*' + k + r':\s*(\d+)', tail)
    info[k] = mm.group(1) if mm else None
print(m.group(1), info)


def kind(k):
    if k.startswith('v_mfma'):
        return 'mfma'
    if k.startswith('v_'):
        return 'valu'
    if k.startswith('ds_'):
        return 'lds'
    if k.startswith(('global', 'buffer', 'flat')):
        return 'vmem'
    if k.startswith('scratch'):
        return 'scratch'
    if k.startswith('s_waitcnt'):
        return 'wait'
    if k.startswith('s_nop'):
        return 'nop'
    return 'salu'


blocks = re.split(r'\n(\.LBB\d+_\d+):', body)
for i in range(1, len(blocks) - 1, 2):
    b = blocks[i + 1]
    ins = [l.split()[0] for l in b.split('\n') if l.startswith('\t') and l.split() and not l.strip().startswith(('.', ';'))]
    c = Counter(kind(k) for k in ins)
    if c['mfma'] >= min_mfma or c['scratch']:
        waits = [l.strip().replace('s_waitcnt ', '') for l in b.split('\n') if 's_waitcnt' in l]
        print('  ', blocks[i], len(ins), dict(c), waits[:24])
