import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _gpu_available():
    try:
        from tenstream_amd import _lib

        return _lib.load().tsx_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the HIP path: a missing library/device is a failure, not a skip."""
    from tenstream_amd import _lib

    lib = _lib.load()
    assert lib.tsx_device_count() > 0, "no HIP device visible: -m gpu tests need the MI355X box"
    return lib
