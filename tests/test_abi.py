"""CPU-side checks of the C-ABI boundary: the library loads, exports everything include/acdsp.h
declares, and refuses to compute without a gfx950 device (no fallback path)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "acdsp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(acdsp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ac_dsp_amd._lib as L
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L.lib, n), "libacdsp.so does not export %s" % n
    assert set(names) == set(L.SYMBOLS), set(names) ^ set(L.SYMBOLS)
    assert L.lib.acdsp_abi_version() == 1


def test_descriptor_layout_matches_header():
    import ac_dsp_amd._lib as L
    assert C.sizeof(L.Fmt) == 20
    assert C.sizeof(L.FirDesc) == 5 * 4 + 4 * 20 + 2 * 4
    assert C.sizeof(L.CicDesc) == 5 * 4 + 2 * 20 + 2 * 4
    assert C.sizeof(L.PolyDecDesc) == 3 * 4 + 4 * 20 + 2 * 4
    assert C.sizeof(L.PolyIntrDesc) == 5 * 4 + 4 * 20 + 2 * 4
    assert C.sizeof(L.IntgDumpDesc) == 3 * 4 + 3 * 20 + 2 * 4
    assert C.sizeof(L.StreamHdr) == 64
    assert [L.lib.acdsp_elem_bytes(w) for w in (1, 16, 17, 32, 33, 64)] == [2, 2, 4, 4, 8, 8]


def test_raw_integer_stream_files_round_trip(tmp_path):
    # host-side file format (no device involved): 64-byte header + [channel][stride] containers
    import numpy as np
    import ac_dsp_amd as A
    rng = np.random.default_rng(1)
    for fmt, dt in ((A.Fmt(16, 2), np.int16), (A.Fmt(36, 21), np.int64), (A.Fmt(24, 9, True, "RND", "SAT"), np.int32)):
        x = rng.integers(-(1 << (fmt.W - 1)), 1 << (fmt.W - 1), size=(5, 37))
        f = tmp_path / ("s%d.acdspraw" % fmt.W)
        A.save_stream(f, x, fmt)
        raw = open(f, "rb").read()
        assert raw[:8] == b"ACDSPRAW" and len(raw) == 64 + 5 * 37 * np.dtype(dt).itemsize
        y, g = A.load_stream(f)
        assert np.array_equal(x, y) and (g.W, g.I, g.S, g.Q, g.O) == (fmt.W, fmt.I, fmt.S, fmt.Q, fmt.O)
    bad = tmp_path / "bad"
    bad.write_bytes(b"not a stream at all, just sixty-four bytes of something else....")
    with pytest.raises(A.AcdspError):
        A.load_stream(bad)
    with pytest.raises(A.AcdspError):
        A.load_stream(tmp_path / "missing")


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ac_dsp_amd as A
    assert A.device_count() == 0
    with pytest.raises(A.AcdspError) as e:
        A.Fir(27, "FOLD_ODD", A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2))
    assert e.value.code == 4  # ACDSP_ENODEVICE
    with pytest.raises(A.AcdspError):
        A.Cic(False, 8, 1, 5, A.Fmt(32, 16), A.Fmt(47, 31))


def test_argument_validation_happens_before_device_use():
    import ac_dsp_amd as A
    with pytest.raises(A.AcdspError) as e:
        A.Fir(8, "FOLD_ODD_ANTI", A.Fmt(16, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2))
    assert e.value.code == 2  # ACDSP_EUNSUPPORTED, like the reference's unassigned output
    with pytest.raises(A.AcdspError) as e:
        A.Fir(8, "SHIFT_REG", A.Fmt(80, 2), A.Fmt(16, 2), A.Fmt(40, 12), A.Fmt(16, 2))
    assert e.value.code == 2


def test_node_shard_rule_and_no_device_errors():
    """acdsp_node_shard needs no device: contiguous slices that cover the bank, sizes within one of each other, the same rule
    bench.py's process-per-GPU mode uses; creating a node handle without a GPU fails loudly (no CPU path behind it either)."""
    import sys
    sys.path.insert(0, ROOT)
    import ac_dsp_amd as A
    import bench
    for n_total, n in ((8192, 8), (1030, 8), (7, 2), (3, 3), (16384, 5)):
        cuts = [A.node_shard(n_total, n, s) for s in range(n)]
        assert cuts == [bench.shard(n_total, n, s) for s in range(n)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n_total and all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
    with pytest.raises(A.AcdspError):
        A.node_shard(10, 0, 0)
    import torch
    if not torch.cuda.is_available():
        f = A.Fmt(16, 2)
        with pytest.raises(A.AcdspError):
            A.NodeFir(31, "SHIFT_REG", f, f, A.Fmt(40, 12), f, 64, [0, 0])


def test_product_never_imports_the_oracle():
    bad = []
    for base in ("ac_dsp_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle|#include\s*[<\"].*oracle|libacdsp_oracle", txt, flags=re.M):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


# Kernels allowed to carry a private segment, with the reason.  Everything else in libacdsp.so must have none: a spill inside a
# streaming loop is a VMEM operation every counted wait has to drain, and round 3 shipped 39 kernels with scratch.
SCRATCH_ALLOWED = {
    # exact per-MAC poly_intr kernels (128-bit arithmetic, correctness path): 20 bytes RESERVED for the VGPR that holds spilled
    # SGPRs -- the ISA of both kernels contains no scratch instruction (`.uses_flat_scratch, 0`; hipcc -S shows none)
    "polyintr_kernel": 20,
    "polyintr_save_kernel": 20,
    # 255 taps, DENSE 16-bit set, OUT = ACC in int64 containers: 72 VGPRs of Toeplitz fragments + two 48-register accumulator sets +
    # the 64-bit epilogue leave 9 registers too few at two waves per SIMD; the band-limited instantiations <9,3,34|35|50|51,1>
    # (the bench's fir255_wide row among them) are clean
    "fir_mfma_kernel<9, 3, 0, 1, 0, false>": 40,
    # one K-block, 4-byte containers: 36 bytes reserved for the VGPR that holds spilled SGPRs (the argument block grew by the general
    # rounding constants and the unsigned-sample flip); hipcc -S of fir_mfma_alt.hip: 0 scratch instructions in the function
    "fir_mfma_kernel<1, 3, 0, 1, 0, true>": 36,
}


def test_no_kernel_uses_scratch():
    """Parses the code-object notes of the shipped library (tools/codeobj_notes.py: pure Python, msgpack metadata of every
    gfx950 code object) and fails on any kernel with a private segment or spilled VGPRs outside SCRATCH_ALLOWED."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import codeobj_notes as N
    import ac_dsp_amd._lib as L
    ks = N.kernels(L.LIB_PATH)
    assert len(ks) >= 300, "expected the engine's kernels in %s, found %d" % (L.LIB_PATH, len(ks))
    names = N.demangle([k["name"] for k in ks])
    bad = []
    for k, dn in zip(ks, names):
        short = dn.replace("acdsp::", "")
        if k["scratch"] > 0 or k["vgpr_spill"] > 0:
            if SCRATCH_ALLOWED.get(short, -1) >= k["scratch"]:
                continue
            bad.append("%s: scratch %d B, %d VGPRs spilled" % (short, k["scratch"], k["vgpr_spill"]))
    assert not bad, "kernels with scratch memory:\n  " + "\n  ".join(bad)
    # the allow-list must not outlive its entries
    present = {dn.replace("acdsp::", "") for k, dn in zip(ks, names) if k["scratch"] > 0}
    assert set(SCRATCH_ALLOWED) <= present, "stale SCRATCH_ALLOWED entries: %s" % (set(SCRATCH_ALLOWED) - present)
