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


# ---- parity record: near-tie swap counts of the golden search cases -------------------------
# tests/test_search_gpu.py::test_search_golden files, per golden case, how many ranks differed
# from the reference's recorded row order (single-query path and batch path separately).  The
# table is printed in the terminal summary (so it lands in the driver's GPU test log) and written
# to gpurun_out/parity_swaps.json (merged back from the GPU box).
PARITY_SWAPS = {}


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not PARITY_SWAPS:
        return
    import json
    tr = terminalreporter
    tr.write_sep("-", "parity: ranks that differ from the reference's golden row order")
    for name, rec in PARITY_SWAPS.items():
        tr.write_line(f"{name}: single-query path {rec['single']}, batch path {rec['batch']}, "
                      f"coalesced route {rec.get('coalesced')} (min adjacent f64 gap {rec['min_gap']})")
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "parity_swaps.json"), "w") as f:
            json.dump(PARITY_SWAPS, f, indent=1)
    except OSError:
        pass


@pytest.fixture
def first_rows_thresholds():
    """The fused batch path takes its thresholds from the FIRST rows of the corpus for the duration of the test (rounds
    1-3's layout; svs_internal_tune(2, 0)), so that a corpus sorted by similarity to the queries still overflows the
    candidate lists and the fallback paths stay under test.  The default -- a sample spread over the corpus -- is back after."""
    from svs_amd import _native
    lib = _native.load()
    assert lib.svs_internal_tune(2, 0) == 0
    yield
    assert lib.svs_internal_tune(2, 1) == 0
