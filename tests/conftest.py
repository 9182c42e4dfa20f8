import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count() -> int:
    try:
        from svs_amd import _native
        return _native.device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the HIP path: if the library or the device is missing
    they FAIL (no skip, no fallback)."""
    from svs_amd import _native
    _native.load()
    n = _native.device_count()
    assert n > 0, "no HIP device visible: -m gpu tests need a real MI355X"
    return n
