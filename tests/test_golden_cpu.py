"""The CPU oracle against the vectors the reference's own headers produced (tests/golden/ref_hdr, tools/gen_golden):
every class of SURVEY section 8, all FTYPEs, the BASELINE types, lossy / saturating / unsigned accumulators, mid-stream
coefficient reloads and chunked calls.  This is what pins oracle/acdsp_oracle.c to the reference's loop source."""
import numpy as np
import pytest

import golden_cases as G
from oracle import Fmt, OracleFir, OracleCic, OraclePolyDec, OraclePolyIntr, OracleIntgDump, OracleMvAvg

FIR = G.load(("const", "load", "prog"))
RS = G.load(("reg_share",))
CIC = G.load(("cic_dec", "cic_intr"))
PDEC = G.load(("poly_dec",))
PINT = G.load(("poly_intr",))
IDMP = G.load(("intg_dump",))
MVA = G.load(("mv_avg",))


def F(a):
    return G.fmt_of(Fmt, a)


def test_fixture_inventory():
    names = [c["name"] for c in FIR]
    for ft in ("SHIFT_REG", "ROTATE_SHIFT", "C_BUFF", "FOLD_EVEN", "FOLD_ODD", "TRANSPOSED"):
        for cls in ("const", "load", "prog"):
            assert any(n.startswith("%s_base_%s_255" % (cls, ft)) for n in names), (cls, ft)
    # BASELINE configs[0] as SURVEY 8(d) fixes it: 63 taps <16,2,true>, ACC = OUT <38,10>, the testbench's two-tone stimulus, 1024 samples
    cfg1 = [c for c in FIR if c["name"].startswith("const_cfg1_63_")]
    assert len(cfg1) == 6 and all(c["n_taps"] == 63 and c["acc"][:3] == [38, 10, 1] and c["out"][:3] == [38, 10, 1] and len(c["x"]) == 1024 for c in cfg1)
    assert len(FIR) >= 100 and len(RS) >= 10 and len(CIC) >= 12 and len(PDEC) >= 5 and len(PINT) >= 8 and len(IDMP) >= 4


@pytest.mark.parametrize("c", FIR, ids=G.ids(FIR))
def test_fir_oracle_matches_reference_headers(c):
    o = OracleFir(c["n_taps"], c["ftype"], F(c["in"]), F(c["coeff"]), F(c["acc"]), F(c["out"]))
    x = G.arr(c, "x")
    y = np.concatenate([o.run(cf, x[a:b])[0] for cf, a, b in G.segments(c)])
    assert np.array_equal(y, G.arr(c, "y"))


@pytest.mark.parametrize("c", RS, ids=G.ids(RS))
def test_reg_share_oracle_matches_reference_header(c):
    o = OracleFir(c["n_taps"], c["ftype"], F(c["in"]), F(c["coeff"]), F(c["acc"]), F(c["out"]),
                  reg_share=(c["mem_word_width"], c["blk_sz"], c["blk_offset"]))
    x = G.arr(c, "x")
    y, dl = [], []
    for t in range(len(x)):
        y.append(o.run(G.arr(c, "coeffs"), x[t:t + 1])[0, 0])
        dl.append(o.delay_line()[0])
    assert np.array_equal(np.array(y), G.arr(c, "y"))
    assert np.array_equal(np.array(dl), G.arr(c, "delay_line"))


@pytest.mark.parametrize("c", CIC, ids=G.ids(CIC))
def test_cic_oracle_matches_reference_headers(c):
    o = OracleCic(c["class"] == "cic_intr", c["R"], c["M"], c["N"], F(c["in"]), F(c["out"]))
    x = G.arr(c, "x")
    pos, ys = 0, []
    for k, want in zip(c["calls"], c["outs_per_call"]):
        yk = o.run(x[pos:pos + k])[0]
        assert len(yk) == want, "call of %d inputs produced %d outputs, the reference %d" % (k, len(yk), want)
        ys.append(yk)
        pos += k
    assert np.array_equal(np.concatenate(ys), G.arr(c, "y"))


@pytest.mark.parametrize("c", PDEC, ids=G.ids(PDEC))
def test_poly_dec_oracle_matches_reference_header(c):
    o = OraclePolyDec(c["n_taps"], c["df"], F(c["in"]), F(c["coeff"]), F(c["acc"]), F(c["out"]))
    x = G.arr(c, "x")
    pos, ys = 0, []
    for k in c["calls"]:
        ys.append(o.run(G.arr(c, "coeffs"), x[pos:pos + k])[0])
        pos += k
    assert np.array_equal(np.concatenate(ys), G.arr(c, "y"))


@pytest.mark.parametrize("c", PINT, ids=G.ids(PINT))
def test_poly_intr_oracle_matches_reference_header(c):
    o = OraclePolyIntr(c["n_taps"], c["coeff_sz"], c["ifac"], c["ftype"], F(c["in"]), F(c["coeff"]), F(c["acc"]), F(c["out"]))
    x = G.arr(c, "x")
    ra = c["reload_at"]
    if ra < 0:
        y = o.run(c["coeffs"], c["sign"], c["corr"], x)[0]
    else:
        y = np.concatenate([o.run(c["coeffs"], c["sign"], c["corr"], x[:ra])[0], o.run(c["coeffs2"], c["sign2"], c["corr2"], x[ra:])[0]])
    assert sum(c["outs_per_sample"]) == len(c["y"])
    assert np.array_equal(y, G.arr(c, "y"))


@pytest.mark.parametrize("c", IDMP, ids=G.ids(IDMP))
def test_intg_dump_oracle_matches_reference_header(c):
    o = OracleIntgDump(c["ns"], c["chn"], F(c["in"]), F(c["acc"]), F(c["out"]))
    x, ns = G.arr(c, "x"), G.arr(c, "n_sample")
    xp, bp, ys = 0, 0, []
    for nb in c["blocks_per_call"]:
        blk = ns[bp:bp + nb]
        need = int(sum((v if 1 <= v <= c["ns"] else c["ns"]) for v in blk)) * c["chn"]
        ys.append(o.run(x[xp:xp + need], blk)[0])
        xp += need
        bp += nb
    assert xp == len(x)
    assert np.array_equal(np.concatenate(ys), G.arr(c, "y"))


@pytest.mark.parametrize("c", MVA, ids=G.ids(MVA))
def test_mv_avg_oracle_matches_reference_header(c):
    """The reference's run() / mvAvgCore() source over include/ac_types/ac_window.h (index arithmetic) against the oracle's
    register-and-flags window: frame loop, flush iterations, ACC cast and MAC order are the reference's; the boundary rules
    are this repo's reading of ac_window_1d_flag on both sides (parity unpinned for those)."""
    o = OracleMvAvg(c["taps"], c["win_mode"], F(c["in"]), F(c["coeff"]), F(c["acc"]), F(c["out"]))
    y = o.run(G.arr(c, "coeffs"), G.arr(c, "x"), c["n_sample"])[0]
    assert np.array_equal(y, G.arr(c, "y"))


def test_mv_avg_known_answers():
    """Hand-derived: x = 1..10, c = [1, 2, 4, 8, 16] (oldest sample first), exact types."""
    f, a = Fmt(16, 8), Fmt(32, 16)
    c = np.array([1, 2, 4, 8, 16], dtype=np.int64) * 256
    x = np.arange(1, 11, dtype=np.int64) * 256
    want = {"WIN": [129, 160, 191, 222, 253, 284],
            "MIRROR": [75, 100, 129, 160, 191, 222, 253, 284, 283, 266],     # x[-k] = x[k], x[9+k] = x[9-k]
            "CLIP": [71, 99, 129, 160, 191, 222, 253, 284, 299, 306]}        # edge sample replicated
    for mode, w in want.items():
        y = OracleMvAvg(5, mode, f, f, a, a).run(c, x, 10)[0]
        assert (y // 65536).tolist() == w and not (y % 65536).any(), mode
        y2 = OracleMvAvg(5, mode, f, f, a, a).run(c, np.concatenate([x, x]), 10)[0]       # frames are independent
        assert np.array_equal(y2, np.concatenate([y, y]))
    # frames shorter than the window: the mirror bounces, the clip repeats
    assert (OracleMvAvg(5, "MIRROR", f, f, a, a).run(c, x[:2], 2)[0] // 65536).tolist() == [41, 52]
    assert (OracleMvAvg(5, "CLIP", f, f, a, a).run(c, x[:2], 2)[0] // 65536).tolist() == [55, 59]
    assert (OracleMvAvg(5, "MIRROR", f, f, a, a).run(c, x[:1], 1)[0] // 65536).tolist() == [31]
    assert OracleMvAvg(5, "WIN", f, f, a, a).run(c, x[:4], 4).shape == (1, 0)
    with pytest.raises(ValueError):
        OracleMvAvg(4, "WIN", f, f, a, a).run(c[:4], x, 10)                               # even TAPS: coeffs[TAPS] is read
