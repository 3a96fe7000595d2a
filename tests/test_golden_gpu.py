"""The HIP engine (through the C ABI) against the vectors the reference's own headers produced (tests/golden/ref_hdr,
tools/gen_golden): the third leg of the parity argument -- reference loop source -> vectors <- engine, with the oracle
pinned on the same vectors by tests/test_golden_cpu.py.  Every case runs as several identical channels so that the
batched kernels (matrix-core paths included) see more than one row."""
import numpy as np
import pytest
import torch

import ac_dsp_amd as A
import golden_cases as G

pytestmark = pytest.mark.gpu

FIR = G.load(("const", "load", "prog"))
RS = G.load(("reg_share",))
CIC = G.load(("cic_dec", "cic_intr"))
PDEC = G.load(("poly_dec",))
PINT = G.load(("poly_intr",))
IDMP = G.load(("intg_dump",))
MVA = G.load(("mv_avg",))
NCH = 3


def F(a):
    return G.fmt_of(A.Fmt, a)


def dev(x, fmt):
    """1-D raw words -> [NCH][n] device tensor of fmt's containers (identical rows)."""
    return torch.from_numpy(np.tile(np.asarray(x, dtype=np.int64), (NCH, 1))).to(A.torch_dtype_for(fmt)).cuda()


def host(t):
    y = t.cpu().numpy().astype(np.int64)
    assert all(np.array_equal(y[0], y[i]) for i in range(1, y.shape[0])), "identical channels gave different rows"
    return y[0]


@pytest.mark.parametrize("force_generic", [False, True], ids=["fast", "generic"])
@pytest.mark.parametrize("c", FIR, ids=G.ids(FIR))
def test_fir_engine_matches_reference_headers(c, force_generic):
    fin, fo = F(c["in"]), F(c["out"])
    fir = A.Fir(c["n_taps"], c["ftype"], fin, F(c["coeff"]), F(c["acc"]), fo, n_channels=NCH, kind=c["class"], force_generic=force_generic)
    x = G.arr(c, "x")
    ys, last = [], None
    for cf, a, b in G.segments(c):
        if last is None or not np.array_equal(cf, last):
            fir.set_coeffs(cf)
            last = cf
        ys.append(host(fir.run(dev(x[a:b], fin))))
    y = np.concatenate(ys)
    want = G.arr(c, "y")
    bad = np.nonzero(y != want)[0]
    assert bad.size == 0, "%d mismatches, first at %d: got %d want %d (path %s)" % (len(bad), bad[0], y[bad[0]], want[bad[0]], fir.path)


def test_baseline_types_take_the_matrix_core_path():
    c = next(c for c in FIR if c["name"] == "const_base_SHIFT_REG_255")
    fir = A.Fir(255, "SHIFT_REG", F(c["in"]), F(c["coeff"]), F(c["acc"]), F(c["out"]), n_channels=NCH, kind="const")
    fir.set_coeffs(G.arr(c, "coeffs"))
    assert fir.path == "mfma_i8"


@pytest.mark.parametrize("c", RS, ids=G.ids(RS))
def test_reg_share_engine_matches_reference_header(c):
    fin = F(c["in"])
    fir = A.Fir(c["n_taps"], c["ftype"], fin, F(c["coeff"]), F(c["acc"]), F(c["out"]), n_channels=NCH, kind="reg_share")
    fir.set_coeffs(G.reg_share_taps(c))
    x = G.arr(c, "x")
    y = np.concatenate([host(fir.run(dev(x[a:b], fin))) for a, b in ((0, 7), (7, len(x)))])
    assert np.array_equal(y, G.arr(c, "y"))


@pytest.mark.parametrize("c", CIC, ids=G.ids(CIC))
def test_cic_engine_matches_reference_headers(c):
    fin = F(c["in"])
    cic = A.Cic(c["class"] == "cic_intr", c["R"], c["M"], c["N"], fin, F(c["out"]), n_channels=NCH)
    x = G.arr(c, "x")
    pos, ys = 0, []
    for k, want in zip(c["calls"], c["outs_per_call"]):
        yk = cic.run(dev(x[pos:pos + k], fin))
        assert yk.shape[1] == want, "call of %d inputs produced %d outputs, the reference %d" % (k, yk.shape[1], want)
        if want:
            ys.append(host(yk))
        pos += k
    assert np.array_equal(np.concatenate(ys), G.arr(c, "y"))


@pytest.mark.parametrize("c", PDEC, ids=G.ids(PDEC))
def test_poly_dec_engine_matches_reference_header(c):
    fin = F(c["in"])
    pd = A.PolyDec(c["n_taps"], c["df"], fin, F(c["coeff"]), F(c["acc"]), F(c["out"]), n_channels=NCH)
    pd.set_coeffs(G.arr(c, "coeffs"))
    x = G.arr(c, "x")
    pos, ys = 0, []
    for k in c["calls"]:
        ys.append(host(pd.run(dev(x[pos:pos + k], fin))))
        pos += k
    assert np.array_equal(np.concatenate(ys), G.arr(c, "y"))


@pytest.mark.parametrize("c", PINT, ids=G.ids(PINT))
def test_poly_intr_engine_matches_reference_header(c):
    fin = F(c["in"])
    pi = A.PolyIntr(c["n_taps"], c["coeff_sz"], c["ifac"], c["ftype"], fin, F(c["coeff"]), F(c["acc"]), F(c["out"]), n_channels=NCH)
    pi.set_ctrl(c["coeffs"], c["sign"], c["corr"])
    x = G.arr(c, "x")
    ra = c["reload_at"]
    if ra < 0:
        y = host(pi.run(dev(x, fin)))
    else:
        y0 = host(pi.run(dev(x[:ra], fin)))
        pi.set_ctrl(c["coeffs2"], c["sign2"], c["corr2"])
        y = np.concatenate([y0, host(pi.run(dev(x[ra:], fin)))])
    assert np.array_equal(y, G.arr(c, "y"))


@pytest.mark.parametrize("c", IDMP, ids=G.ids(IDMP))
def test_intg_dump_engine_matches_reference_header(c):
    fin = F(c["in"])
    eng = A.IntgDump(c["ns"], c["chn"], fin, F(c["acc"]), F(c["out"]), n_objects=NCH)
    x, ns = G.arr(c, "x"), G.arr(c, "n_sample")
    xp, bp, ys = 0, 0, []
    for nb in c["blocks_per_call"]:
        blk = ns[bp:bp + nb]
        need = int(sum((v if 1 <= v <= c["ns"] else c["ns"]) for v in blk)) * c["chn"]
        yk = eng.run(dev(x[xp:xp + need], fin), blk)
        if yk.shape[1]:
            ys.append(host(yk))
        xp += need
        bp += nb
    got = np.concatenate(ys) if ys else np.zeros(0, dtype=np.int64)
    assert np.array_equal(got, G.arr(c, "y"))


@pytest.mark.parametrize("force_generic", [False, True], ids=["fast", "generic"])
@pytest.mark.parametrize("c", MVA, ids=G.ids(MVA))
def test_mv_avg_engine_matches_reference_header(c, force_generic):
    fin = F(c["in"])
    eng = A.MvAvg(c["max_sample"], c["taps"], c["win_mode"], fin, F(c["coeff"]), F(c["acc"]), F(c["out"]), n_objects=NCH, force_generic=force_generic)
    eng.set_coeffs(G.arr(c, "coeffs"))
    y = eng.run(dev(G.arr(c, "x"), fin), c["n_sample"])
    want = G.arr(c, "y")
    assert y.shape[1] == len(want)
    if len(want):
        assert np.array_equal(host(y), want)


NOREL = [c for c in FIR if c.get("reload_at", -1) < 0]


@pytest.mark.parametrize("c", NOREL, ids=G.ids(NOREL))
def test_fir_engine_matches_reference_headers_on_distinct_channels(c):
    """The same vectors with a DIFFERENT stream in every channel: channel k carries the fixture delayed by d_k samples (zeros in
    front).  The reference objects start from zeroed registers (ac_fir_const_coeffs.h:145), so channel k must produce the
    fixture's output delayed by d_k -- which checks per-row addressing, history and call splitting against reference-made
    numbers, not only against the oracle (the identical-channel test above cannot see a row mix-up)."""
    fin, fo = F(c["in"]), F(c["out"])
    x, want = G.arr(c, "x"), G.arr(c, "y")
    n = len(x)
    delays = [0, 1 + (n // 7), 3 + (n // 3)]
    rows = np.zeros((len(delays), n), dtype=np.int64)
    for k, d in enumerate(delays):
        rows[k, d:] = x[:n - d]
    fir = A.Fir(c["n_taps"], c["ftype"], fin, F(c["coeff"]), F(c["acc"]), fo, n_channels=len(delays), kind=c["class"])
    fir.set_coeffs(G.arr(c, "coeffs"))
    xd = torch.from_numpy(rows).to(A.torch_dtype_for(fin)).cuda()
    ys = [fir.run(xd[:, a:b].contiguous()).cpu().numpy().astype(np.int64) for _, a, b in G.segments(c)]
    y = np.concatenate(ys, axis=1)
    for k, d in enumerate(delays):
        assert np.array_equal(y[k, d:], want[:n - d]) and not y[k, :d].any(), "channel %d (delay %d), path %s" % (k, d, fir.path)


DEC = [c for c in CIC if c["class"] == "cic_dec"]


@pytest.mark.parametrize("c", DEC, ids=G.ids(DEC))
def test_cic_decimator_matches_reference_headers_on_distinct_channels(c):
    """As above for ac_cic_dec_full: delays are multiples of R, so the decimation phase of every channel is the fixture's."""
    fin, fo = F(c["in"]), F(c["out"])
    x, want = G.arr(c, "x"), G.arr(c, "y")
    n, R = len(x), c["R"]
    delays = [0, R * (1 + n // (9 * R)), R * (2 + n // (4 * R))]
    rows = np.zeros((len(delays), n), dtype=np.int64)
    for k, d in enumerate(delays):
        rows[k, d:] = x[:n - d]
    cic = A.Cic(False, R, c["M"], c["N"], fin, fo, n_channels=len(delays))
    xd = torch.from_numpy(rows).to(A.torch_dtype_for(fin)).cuda()
    pos, ys = 0, []
    for k in c["calls"]:
        ys.append(cic.run(xd[:, pos:pos + k].contiguous()).cpu().numpy().astype(np.int64))
        pos += k
    y = np.concatenate(ys, axis=1)
    for k, d in enumerate(delays):
        m = d // R
        assert np.array_equal(y[k, m:], want[:y.shape[1] - m]) and not y[k, :m].any(), "channel %d (delay %d)" % (k, d)


# ---- round 4: a DIFFERENT stream in every row for the remaining families (a row mix-up is invisible to identical rows) ----

REL = [c for c in FIR if c.get("reload_at", -1) >= 0]


@pytest.mark.parametrize("c", REL, ids=G.ids(REL))
def test_fir_reload_cases_on_distinct_channels(c):
    """Mid-stream coefficient reloads (ac_fir_load_coeffs / ac_fir_prog_coeffs; TRANSPOSED differs there) with channel k carrying the
    fixture delayed by d_k samples AND reloading d_k samples later: a bank of per-channel coefficient sets, each channel switching
    to the fixture's second set at reload_at + d_k.  A zeroed filter fed zeros stays zeroed, so channel k must reproduce the
    fixture's output delayed by d_k."""
    fin, fo = F(c["in"]), F(c["out"])
    x, want = G.arr(c, "x"), G.arr(c, "y")
    n, ra = len(x), c["reload_at"]
    delays = [0, 1 + n // 7, 2 + n // 4]
    rows = np.zeros((len(delays), n), dtype=np.int64)
    for k, d in enumerate(delays):
        rows[k, d:] = x[:n - d]
    cuts = [0]
    for k in c["calls"]:
        cuts.append(cuts[-1] + k)
    cuts = sorted(set(cuts + [min(n, ra + d) for d in delays]))
    fir = A.Fir(c["n_taps"], c["ftype"], fin, F(c["coeff"]), F(c["acc"]), fo, n_channels=len(delays), kind=c["class"], coeffs_per_channel=True)
    xd = torch.from_numpy(rows).to(A.torch_dtype_for(fin)).cuda()
    c1, c2 = G.arr(c, "coeffs"), G.arr(c, "coeffs2")
    ys, last = [], None
    for a, b in zip(cuts[:-1], cuts[1:]):
        sets = np.stack([c2 if a >= ra + d else c1 for d in delays])
        if last is None or not np.array_equal(sets, last):
            fir.set_coeffs(sets)
            last = sets
        ys.append(fir.run(xd[:, a:b].contiguous()).cpu().numpy().astype(np.int64))
    y = np.concatenate(ys, axis=1)
    for k, d in enumerate(delays):
        assert np.array_equal(y[k, d:], want[:n - d]) and not y[k, :d].any(), "channel %d (delay %d), path %s" % (k, d, fir.path)


@pytest.mark.parametrize("c", RS, ids=G.ids(RS))
def test_reg_share_matches_reference_header_on_distinct_channels(c):
    fin = F(c["in"])
    x, want = G.arr(c, "x"), G.arr(c, "y")
    n = len(x)
    delays = [0, 1 + n // 7, 3 + n // 3]
    rows = np.zeros((len(delays), n), dtype=np.int64)
    for k, d in enumerate(delays):
        rows[k, d:] = x[:n - d]
    fir = A.Fir(c["n_taps"], c["ftype"], fin, F(c["coeff"]), F(c["acc"]), F(c["out"]), n_channels=len(delays), kind="reg_share")
    fir.set_coeffs(G.reg_share_taps(c))
    xd = torch.from_numpy(rows).to(A.torch_dtype_for(fin)).cuda()
    y = np.concatenate([fir.run(xd[:, a:b].contiguous()).cpu().numpy().astype(np.int64) for a, b in ((0, 7), (7, n))], axis=1)
    for k, d in enumerate(delays):
        assert np.array_equal(y[k, d:], want[:n - d]) and not y[k, :d].any(), "channel %d (delay %d)" % (k, d)


@pytest.mark.parametrize("c", PDEC, ids=G.ids(PDEC))
def test_poly_dec_matches_reference_header_on_distinct_channels(c):
    """Delays are multiples of DF, so every channel keeps the fixture's decimation phase."""
    fin = F(c["in"])
    x, want = G.arr(c, "x"), G.arr(c, "y")
    n, df = len(x), c["df"]
    delays = [0, df * (1 + n // (9 * df)), df * (2 + n // (4 * df))]
    rows = np.zeros((len(delays), n), dtype=np.int64)
    for k, d in enumerate(delays):
        rows[k, d:] = x[:n - d]
    pd = A.PolyDec(c["n_taps"], df, fin, F(c["coeff"]), F(c["acc"]), F(c["out"]), n_channels=len(delays))
    pd.set_coeffs(G.arr(c, "coeffs"))
    xd = torch.from_numpy(rows).to(A.torch_dtype_for(fin)).cuda()
    pos, ys = 0, []
    for k in c["calls"]:
        ys.append(pd.run(xd[:, pos:pos + k].contiguous()).cpu().numpy().astype(np.int64))
        pos += k
    y = np.concatenate(ys, axis=1)
    for k, d in enumerate(delays):
        m = d // df
        assert np.array_equal(y[k, m:], want[:y.shape[1] - m]) and not y[k, :m].any(), "channel %d (delay %d)" % (k, d)


PINT_NOREL = [c for c in PINT if c["reload_at"] < 0]


@pytest.mark.parametrize("c", PINT_NOREL, ids=G.ids(PINT_NOREL))
def test_poly_intr_matches_reference_header_on_distinct_channels(c):
    """Channel k = the fixture delayed by d_k input samples: its outputs are the fixture's delayed by d_k * IF (the folded cores emit
    the sums of sample i - 1 when sample i arrives, ac_poly_intr.h:153-175, so the first sample of the STREAM emits nothing and a
    leading zero sample emits IF zeros).  Control reloads are shared by the channels of a handle: those cases keep identical rows."""
    fin = F(c["in"])
    x, want = G.arr(c, "x"), G.arr(c, "y")
    n, IF = len(x), c["ifac"]
    delays = [0, 2, 5 + n // 6]
    rows = np.zeros((len(delays), n), dtype=np.int64)
    for k, d in enumerate(delays):
        rows[k, d:] = x[:n - d]
    pi = A.PolyIntr(c["n_taps"], c["coeff_sz"], IF, c["ftype"], fin, F(c["coeff"]), F(c["acc"]), F(c["out"]), n_channels=len(delays))
    pi.set_ctrl(c["coeffs"], c["sign"], c["corr"])
    xd = torch.from_numpy(rows).to(A.torch_dtype_for(fin)).cuda()
    cut = n // 3
    y = np.concatenate([pi.run(xd[:, :cut].contiguous()).cpu().numpy().astype(np.int64), pi.run(xd[:, cut:].contiguous()).cpu().numpy().astype(np.int64)], axis=1)
    assert y.shape[1] == len(want)
    for k, d in enumerate(delays):
        m = d * IF
        assert np.array_equal(y[k, m:], want[:len(want) - m]) and not y[k, :m].any(), "channel %d (delay %d)" % (k, d)


IDMP_MULTI = [c for c in IDMP if c["chn"] > 1]


@pytest.mark.parametrize("c", IDMP_MULTI, ids=G.ids(IDMP_MULTI))
def test_intg_dump_matches_reference_header_on_distinct_objects(c):
    """Object k carries the fixture with its CHN interleaved channels rotated by k: channels are independent and share the dump
    schedule (ac_intg_dump.h:93-147), so its outputs are the fixture's with the channels rotated likewise."""
    fin, chn = F(c["in"]), c["chn"]
    x, ns, want = G.arr(c, "x"), G.arr(c, "n_sample"), G.arr(c, "y")
    n_obj = chn
    rows = np.stack([np.roll(x.reshape(-1, chn), -k, axis=1).ravel() for k in range(n_obj)])
    eng = A.IntgDump(c["ns"], chn, fin, F(c["acc"]), F(c["out"]), n_objects=n_obj)
    xd = torch.from_numpy(rows).to(A.torch_dtype_for(fin)).cuda()
    xp, bp, ys = 0, 0, []
    for nb in c["blocks_per_call"]:
        blk = ns[bp:bp + nb]
        need = int(sum((v if 1 <= v <= c["ns"] else c["ns"]) for v in blk)) * chn
        yk = eng.run(xd[:, xp:xp + need].contiguous(), blk)
        if yk.shape[1]:
            ys.append(yk.cpu().numpy().astype(np.int64))
        xp += need
        bp += nb
    y = np.concatenate(ys, axis=1)
    for k in range(n_obj):
        assert np.array_equal(y[k], np.roll(want.reshape(-1, chn), -k, axis=1).ravel()), "object %d" % k


MVA_MULTI = [c for c in MVA if c["n_frames"] > 1 and len(c["y"])]


@pytest.mark.parametrize("force_generic", [False, True], ids=["fast", "generic"])
@pytest.mark.parametrize("c", MVA_MULTI, ids=G.ids(MVA_MULTI))
def test_mv_avg_matches_reference_header_on_distinct_objects(c, force_generic):
    """Object k carries the fixture's frames rotated by k (frames are independent windows, ac_mv_avg.h:146-190): its output frames
    are the fixture's rotated likewise."""
    fin, nf = F(c["in"]), c["n_frames"]
    x, want = G.arr(c, "x").reshape(nf, -1), G.arr(c, "y").reshape(nf, -1)
    n_obj = min(nf, 3)
    rows = np.stack([np.roll(x, k, axis=0).ravel() for k in range(n_obj)])
    eng = A.MvAvg(c["max_sample"], c["taps"], c["win_mode"], fin, F(c["coeff"]), F(c["acc"]), F(c["out"]), n_objects=n_obj, force_generic=force_generic)
    eng.set_coeffs(G.arr(c, "coeffs"))
    y = eng.run(torch.from_numpy(rows).to(A.torch_dtype_for(fin)).cuda(), c["n_sample"]).cpu().numpy().astype(np.int64)
    for k in range(n_obj):
        assert np.array_equal(y[k], np.roll(want, k, axis=0).ravel()), "object %d" % k
