"""GPU tests of the C++ boundary: the drop-in class templates (include/ac_dsp) driven by C++ testbenches.

tb_*.cpp are this repo's own testbenches (same stimulus, vectors and pass criteria as the reference's
tests/rtest_*.cpp).  rtest_* are the reference's testbenches themselves, compiled UNCHANGED against our
headers by __graft_entry__.build() in the build container (binaries only travel; the reference sources
never enter the repo); they are run when present."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_bin")
VEC = os.path.join(ROOT, "tests", "golden", "ref_txt")   # the testbenches open their vectors by bare name


def run(exe, env=None):
    e = dict(os.environ)
    e.pop("ACDSP_HOST_SMALL_MACS", None)
    e.update(env or {})
    p = subprocess.run([exe], cwd=VEC, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    return p.returncode, p.stdout.decode(errors="replace")


def kernel_runs(out):
    """ACDSP_TRACE=1 lines of the C-ABI layer: (handles created, run() calls that launched kernels, summed over the destroyed handles)"""
    created = len(re.findall(r"^\[acdsp\] (?:fir|cic)_create ", out, flags=re.M))
    runs = sum(int(v) for v in re.findall(r"^\[acdsp\] (?:fir|cic)_destroy kernel_runs=(\d+)", out, flags=re.M))
    return created, runs


TBS = ("tb_tiny", "tb_fir", "tb_cic", "tb_batched", "tb_polydec", "tb_reg_share", "tb_poly_intr", "tb_intg_dump", "tb_mv_avg", "tb_wide", "tb_node")
# tests/_bin arrived prebuilt (the snapshot of a build container that ran __graft_entry__.build()): then the reference's rtest_*
# binaries must have arrived with it -- a lost binary is a failure there, not a skip
PREBUILT = all(os.path.exists(os.path.join(BIN, t)) for t in TBS)


@pytest.fixture(scope="module", autouse=True)
def built():
    if not all(os.path.exists(os.path.join(BIN, t)) for t in TBS):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])


@pytest.mark.parametrize("tb", list(TBS))
def test_own_cpp_testbench(tb):
    rc, out = run(os.path.join(BIN, tb))
    assert rc == 0 and "Test PASSED." in out, out


# The drop-in FIR classes keep one-channel calls of fewer than ACDSP_HOST_SMALL_MACS samples x taps (default 8192) in the header's own
# ac_fixed loop (include/ac_dsp/acdsp_engine.h: run_values_c) -- with the reference's testbench sizes that is every call of
# rtest_ac_fir_prog_coeffs (one sample x 27 taps per call).  Each testbench therefore runs twice: as shipped, and with the host path
# switched off, where it must have created a device handle and launched kernels for its run() calls (ACDSP_TRACE lines).
HOST_LEGS = [pytest.param(None, id="default"), pytest.param("0", id="host_small_macs_0")]


@pytest.mark.parametrize("small_macs", HOST_LEGS)
@pytest.mark.parametrize("name", ["ac_fir_const_coeffs", "ac_fir_load_coeffs", "ac_fir_prog_coeffs", "ac_cic_dec_full",
                                  "ac_cic_intr_full"])
def test_reference_rtest_binary_unchanged(name, small_macs):
    exe = os.path.join(BIN, "rtest_" + name)
    if not os.path.exists(exe):
        assert not PREBUILT, "tests/_bin holds prebuilt testbenches but %s is missing: __graft_entry__.build() did not run with /root/reference" % exe
        pytest.skip("rtest binary not prebuilt (needs /root/reference at build time)")
    env = {"ACDSP_TRACE": "1"}
    if small_macs is not None:
        env["ACDSP_HOST_SMALL_MACS"] = small_macs
    rc, out = run(exe, env)
    assert rc == 0 and "PASSED" in out, out
    created, runs = kernel_runs(out)
    if small_macs == "0" or "cic" in name:
        # every run() of the testbench went through the C ABI into HIP kernels
        assert created >= 1 and runs >= 1, (created, runs, out[-2000:])


@pytest.mark.parametrize("small_macs", HOST_LEGS)
def test_own_fir_testbench_on_both_sides_of_the_host_threshold(small_macs):
    env = {"ACDSP_TRACE": "1"}
    if small_macs is not None:
        env["ACDSP_HOST_SMALL_MACS"] = small_macs
    rc, out = run(os.path.join(BIN, "tb_fir"), env)
    assert rc == 0 and "Test PASSED." in out, out
    created, runs = kernel_runs(out)
    if small_macs == "0":
        assert created >= 1 and runs >= 1, (created, runs)


@pytest.mark.parametrize("tb,asan", [("tb_tiny_san", False), ("tb_tiny_asan", True)])
def test_host_loop_of_the_drop_in_classes_under_sanitizers(tb, asan):
    """tb_tiny (72 configurations: the header's host loop against the kernels, state blobs moving both ways) built with UBSan + checked std::vector
    indexing, and with AddressSanitizer as well: acdsp_engine.h's host_step indexes its rings with % on every tap."""
    env = {"UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"}
    if asan:
        env["ASAN_OPTIONS"] = "detect_leaks=0:protect_shadow_gap=0"      # (the HIP runtime maps the shadow gap)
    rc, out = run(os.path.join(BIN, tb), env)
    if asan and rc != 0 and "Test PASSED." not in out and ("Shadow memory" in out or "ReserveShadowMemoryRange" in out or "failed to" in out.lower() and "shadow" in out.lower()):
        pytest.skip("AddressSanitizer cannot start beside the HIP runtime on this box")
    assert rc == 0 and "Test PASSED." in out and "runtime error" not in out and "AddressSanitizer" not in out, out[-3000:]
