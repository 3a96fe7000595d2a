"""The CPU-side code of the repo under AddressSanitizer + UndefinedBehaviorSanitizer (the reference's memory / race checks have no
counterpart for GPU kernels; the oracle -- the checker everything else is compared with -- and the header-only host loop of the drop-in FIR
classes are the CPU code that can be sanitised).  The oracle's own pin (its tests against the reference's vectors, the golden fixtures and
the known-answer tests) is re-run in a child process on oracle/_san/libacdsp_oracle_san.so."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_pin_under_asan_and_ubsan():
    asan = _libasan()
    if not asan:
        pytest.skip("no libasan in this image")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "sanitize"])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               ACDSP_ORACLE_LIB=os.path.join(ROOT, "oracle", "_san", "libacdsp_oracle_san.so"))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), os.path.join(ROOT, "tests", "test_golden_cpu.py"),
                        os.path.join(ROOT, "tests", "test_wide_cpu.py"), "-q", "-x", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-4000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-4000:]
    assert " passed" in p.stdout
