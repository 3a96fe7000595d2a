"""GPU tests of the C++ boundary: the drop-in class templates (include/ac_dsp) driven by C++ testbenches.

tb_*.cpp are this repo's own testbenches (same stimulus, vectors and pass criteria as the reference's
tests/rtest_*.cpp).  rtest_* are the reference's testbenches themselves, compiled UNCHANGED against our
headers by __graft_entry__.build() in the build container (binaries only travel; the reference sources
never enter the repo); they are run when present."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_bin")
VEC = os.path.join(ROOT, "tests", "golden", "ref_txt")   # the testbenches open their vectors by bare name


def run(exe):
    p = subprocess.run([exe], cwd=VEC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    return p.returncode, p.stdout.decode(errors="replace")


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


@pytest.mark.parametrize("name", ["ac_fir_const_coeffs", "ac_fir_load_coeffs", "ac_fir_prog_coeffs", "ac_cic_dec_full",
                                  "ac_cic_intr_full"])
def test_reference_rtest_binary_unchanged(name):
    exe = os.path.join(BIN, "rtest_" + name)
    if not os.path.exists(exe):
        assert not PREBUILT, "tests/_bin holds prebuilt testbenches but %s is missing: __graft_entry__.build() did not run with /root/reference" % exe
        pytest.skip("rtest binary not prebuilt (needs /root/reference at build time)")
    rc, out = run(exe)
    assert rc == 0 and "PASSED" in out, out
