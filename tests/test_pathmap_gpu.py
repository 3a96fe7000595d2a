"""Which kernel family every FIR descriptor of a fixed grid resolves to (acdsp_fir_kernel_class + the MFMAs issued per 1024 samples),
against the committed table tests/golden/path_map.json (tools/gen_path_map.py).  engine.hip / fir_mfma.hip / fir_gen.hip decide the
family through some twenty hand-written eligibility predicates; a predicate edit that silently re-routes a class is a parity risk (and a
performance cliff) that the differential tests only find by luck.  Here it fails on metadata: no kernel runs."""
import json
import os

import pytest

import pathmap_grid as G

pytestmark = pytest.mark.gpu
TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "path_map.json")


def test_every_descriptor_of_the_grid_resolves_as_committed():
    with open(TABLE) as f:
        want = json.load(f)
    got = {k: G.resolve(a) for k, a in G.grid()}
    assert set(got) == set(want), "grid changed: regenerate tests/golden/path_map.json (tools/gen_path_map.py)"
    diff = ["%s: committed %s, now %s" % (k, want[k], got[k]) for k in sorted(got) if got[k] != want[k]]
    assert not diff, "%d of %d descriptors resolve differently:\n%s" % (len(diff), len(got), "\n".join(diff[:40]))


def test_the_grid_reaches_every_family():
    with open(TABLE) as f:
        fams = {v.split("/")[0] for v in json.load(f).values()}
    assert {"generic", "lossless64", "mfma_i8", "mfma_gen", "wide", "mfma_lossy", "lossy16", "satacc16"} <= fams, fams
