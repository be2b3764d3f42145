import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _default_knobs():
    """{knob: value} right after the library is loaded (before any test turns a knob); {} without the library."""
    try:
        from custom_op_benchmark_amd import _lib
        return _lib.tune_snapshot()
    except Exception:
        return {}


DEFAULT_KNOBS = _default_knobs()


@pytest.fixture(autouse=True)
def _knobs_back_to_default():
    """Safety net: whatever a test did to the tuning knobs, the next one starts from the defaults."""
    yield
    if DEFAULT_KNOBS:
        from custom_op_benchmark_amd import _lib
        _lib.tune_reset()
