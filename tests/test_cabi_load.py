"""The C-ABI library loads and exports every symbol include/rslam.h declares (no compute)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib():
    from ransac_slam_amd import build, api
    build.build()
    return api


def test_exports_every_declared_symbol(hip_lib):
    header = open(os.path.join(ROOT, "include", "rslam.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rslam_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 24
    L = hip_lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"librslam_hip.so does not export {name}"
    # the python binding covers the same set
    assert declared == set(hip_lib.SYMBOLS), declared ^ set(hip_lib.SYMBOLS)


def test_product_library_exports_no_diagnostics(hip_lib):
    """Fault injection, time stamps and value probes (rslam_debug_*) exist in the diagnostic variant only
    (librslam_hip_dbg.so, -DRSLAM_DEBUG); the product library exports exactly the header's entry points."""
    import subprocess
    from ransac_slam_amd import build
    out = subprocess.run(["nm", "-D", "--defined-only", build.LIB], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and "rslam_" in ln.split()[-1] and not ln.split()[-1].startswith("_Z")}
    assert exported == set(hip_lib.SYMBOLS), exported ^ set(hip_lib.SYMBOLS)
    assert not [n for n in exported if "debug" in n]
    dbg = subprocess.run(["nm", "-D", "--defined-only", build.LIB_DEBUG], capture_output=True, text=True, check=True).stdout
    for name in hip_lib.DEBUG_SYMBOLS:
        assert name in dbg, name
    # ... and no behaviour switch is read from the environment of a deployed node
    blob = open(build.LIB, "rb").read()
    for env in (b"RSLAM_SWEEP_EXP", b"RSLAM_SWEEP_STEPS", b"RSLAM_GATE_APART", b"RSLAM_NO_LI_DEFER", b"RSLAM_SWEEP_UNFUSED_K10",
                b"RSLAM_K10_RIDERS_FIRST", b"RSLAM_NO_LI_SMALL", b"RSLAM_LI_SKIP", b"RSLAM_NO_MACRO", b"RSLAM_STAGED_MIN_BLOCKS"):
        assert env not in blob, env


def test_struct_sizes_match_header_layout():
    from ransac_slam_amd.ctypes_defs import Camera, Config, Layout, StageTimes
    assert C.sizeof(Camera) == 7 * 8 + 2 * 4
    assert C.sizeof(Config) == C.sizeof(Camera) + 8 + 8 + 8 + 8 + 4 * 4
    assert C.sizeof(Layout) == 8 + 2 * 8
    assert C.sizeof(StageTimes) == 12 * 8


def test_error_strings_and_version(hip_lib):
    L = hip_lib.lib()
    assert b"gfx950" in L.rslam_version()
    assert L.rslam_error_string(0) == b"ok"
    assert b"no CPU fallback" in L.rslam_error_string(-2)


def test_no_cpu_fallback_without_device(hip_lib):
    """On a box without a GPU the product path must fail loudly, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from ransac_slam_amd import default_config
    with pytest.raises(hip_lib.RslamError) as e:
        hip_lib.RslamHip(default_config())
    assert e.value.code == -2


def test_product_path_never_imports_oracle():
    """The oracle is test infrastructure: nothing under ransac_slam_amd/ may reference it."""
    pkg = os.path.join(ROOT, "ransac_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in src and "rslam_oracle" not in src and "orc_" not in src, f
