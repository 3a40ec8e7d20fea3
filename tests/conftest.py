import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


def _gpu_present():
    """True when a HIP device is visible (no torch involved)."""
    try:
        from safeopt_amd import _hip
        return _hip.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def hip_device():
    """Fail loudly (not skip) when a gpu-marked test runs without the HIP path."""
    from safeopt_amd import _hip
    n = _hip.device_count()
    assert n > 0, "gpu-marked test needs a HIP device and libsafeopt_hip.so"
    return 0


@pytest.fixture(autouse=True)
def _product_backend_after_each_test():
    """CPU tests may install the NumPy stand-in for the grid backend
    (``_oracle_backend.use_oracle_backend``); no test inherits it."""
    yield
    try:
        import safeopt_amd.gp_opt as go
        go._BACKEND_FACTORY = None
    except Exception:
        pass
