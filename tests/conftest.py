import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def gpu():
    """The HIP extension on a live device.  No fallback: a missing .so or device is a hard failure."""
    import coltt_amd
    import ctypes
    L = coltt_amd.lib()
    assert L.coltt_device_count() > 0, "no HIP device visible"
    rc = L.coltt_init(0)
    assert rc == 0, L.coltt_last_error()
    return coltt_amd
