"""The oracle's known-answer tests once more under AddressSanitizer + UBSan (`make -C oracle asan`, SURVEY section 5:
sanitizers run on the CPU build only -- GPU ASan is not available on the pool).  The sanitised library is loaded into a
child interpreter with the ASan runtime preloaded; any report makes the child exit non-zero."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_oracle_kat_under_asan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    lib = os.path.join(ROOT, "oracle", "_build", "librslam_oracle_asan.so")
    assert os.path.exists(lib)
    rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("gcc has no libasan.so runtime here")
    env = dict(os.environ, RSLAM_ORACLE_LIB=lib, LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the known-answer tests and one full frame of each golden fixture through the sanitised build
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_kat.py"),
                        os.path.join(ROOT, "tests", "test_oracle_golden.py") + "::test_oracle_reproduces_golden"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, (r.stdout[-3000:] + r.stderr[-3000:])
    assert "passed" in r.stdout
