import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


class _DebugHip:
    """The diagnostic variant of the product library (librslam_hip_dbg.so, -DRSLAM_DEBUG): same kernels plus the fault
    injection / measurement switches the product build does not contain.  Contexts made here live in that library."""

    def __init__(self, api):
        self.api = api
        self._lib = api.lib(debug=True)

    def RslamHip(self, cfg, device=0):
        return self.api.RslamHip(cfg, device, debug=True)

    def set_sweep_exp(self, mask):
        """RSLAM_SWEEP_EXP switches, process-wide inside the diagnostic library; -1 = back to the environment"""
        rc = self._lib.rslam_debug_set_sweep_exp(int(mask))
        assert rc == 0


@pytest.fixture(scope="session")
def hip_dbg():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (the product path has no CPU fallback)")
    from ransac_slam_amd import api
    return _DebugHip(api)
