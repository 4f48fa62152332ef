import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _row_range_mode_of_the_session():
    """MRLA_TEST_ROW_RANGES=2 pytest -m gpu ...: every test runs with the rows of the channels_last row pipeline cut wherever a map
    has >= 16 rows (mrla_tuning_row_ranges; include/mrla_hip.h) -- the whole suite as a test of the cut kernels.  Tests that pin
    the uncut geometry (the counts of tests/test_host_cpu.py, the x_t-free passes of tests/test_lean_gpu.py, which do not exist
    where rows are cut) are expected to object; everything that compares results must not.  Unset: nothing happens."""
    mode = os.environ.get("MRLA_TEST_ROW_RANGES")
    if mode:
        from mrla_amd import _lib
        _lib.load().mrla_tuning_row_ranges(int(mode))
    yield
