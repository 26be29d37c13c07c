import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
# the suites force kernel variants and test hooks through SCS_AMD_* variables (monkeypatch.setenv): rows of class `ab` / `test` of
# scs_amd/csrc/options.h are reachable from the environment only with this set
os.environ["SCS_AMD_ALLOW_ENV_HOOKS"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun / the driver)")


def _have_gpu():
    try:
        from scs_amd import capi
        lib = capi.load("libscsamd_linsys.so")
        return lib.scs_amd_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def have_gpu():
    return _have_gpu()
