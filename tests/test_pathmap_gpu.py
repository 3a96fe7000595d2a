"""Which kernel family every FIR descriptor of a fixed grid resolves to (acdsp_fir_kernel_class + the MFMAs issued per 1024 samples),
against the committed table tests/golden/path_map.json (tools/gen_path_map.py).  engine.hip / fir_mfma.hip / fir_gen.hip decide the
family through some twenty hand-written eligibility predicates; a predicate edit that silently re-routes a class is a parity risk (and a
performance cliff) that the differential tests only find by luck.  Here it fails on metadata: no kernel runs."""
import json
import os

import pytest

import pathmap_grid as G
import pathmap_ops as GO

pytestmark = pytest.mark.gpu
TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "path_map.json")
TABLE_OPS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "path_map_ops.json")


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


def test_every_cic_polydec_mvavg_intgdump_descriptor_resolves_as_committed():
    """The same net under the other families' dispatch (round 6): ac_cic_dec_full / ac_cic_intr_full over rates x stages x widths x outputs
    (recurrence / one-stage FIR identity / two-stage / wide), ac_poly_dec, ac_mv_avg and ac_intg_dump over their shape predicates -- one small
    aligned call per descriptor, then the path the handle reports (tests/pathmap_ops.py; regenerate with tools/gen_path_map.py)."""
    with open(TABLE_OPS) as f:
        want = json.load(f)
    got = GO.table()
    assert set(got) == set(want), "grid changed: regenerate tests/golden/path_map_ops.json (tools/gen_path_map.py)"
    diff = ["%s: committed %s, now %s" % (k, want[k], got[k]) for k in sorted(got) if got[k] != want[k]]
    assert not diff, "%d of %d descriptors resolve differently:\n%s" % (len(diff), len(got), "\n".join(diff[:40]))


def test_the_ops_grid_reaches_every_cic_family():
    with open(TABLE_OPS) as f:
        t = json.load(f)
    assert {"recurrence", "mfma_gen", "two_stage", "wide"} <= {v for k, v in t.items() if k.startswith("cic|dec")}
    assert {"mfma_gen", "fir_identity", "wide"} <= {v for k, v in t.items() if k.startswith("cic|intr")}
    assert {"stream", "tile", "mfma"} <= {v for k, v in t.items() if k.startswith("intgdump")}
