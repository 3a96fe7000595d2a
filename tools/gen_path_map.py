#!/usr/bin/env python3
"""tools/gen_path_map.py [out.json] -- (descriptor -> kernel family / MFMAs issued per 1024 samples) over the grid of tests/pathmap_grid.py,
from the library as built.  The committed copy is tests/golden/path_map.json; tests/test_pathmap_gpu.py fails when a handle resolves
differently -- an eligibility predicate was edited -- and regenerating the table is then a deliberate, reviewable act."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

torch.zeros(1, device="cuda")   # torch's HIP runtime first: it cannot attach once the library's own has the device
import pathmap_grid as G  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "path_map.json")
table = {k: G.resolve(a) for k, a in G.grid()}
with open(out, "w") as f:
    json.dump(table, f, indent=0, sort_keys=True)
fam = {}
for v in table.values():
    fam[v.split("/")[0]] = fam.get(v.split("/")[0], 0) + 1
print("%d descriptors -> %s" % (len(table), out))
print(json.dumps(fam, sort_keys=True))

# the other families (tests/pathmap_ops.py): one small call per descriptor
import pathmap_ops as GO  # noqa: E402
out2 = os.path.join(os.path.dirname(out), "path_map_ops.json")
t2 = GO.table()
with open(out2, "w") as f:
    json.dump(t2, f, indent=0, sort_keys=True)
fam2 = {}
for k, v in t2.items():
    key = k.split("|")[0] + ("/" + k.split("|")[1] if k.startswith("cic") else "") + ":" + v
    fam2[key] = fam2.get(key, 0) + 1
print("%d descriptors -> %s" % (len(t2), out2))
print(json.dumps(fam2, sort_keys=True))
