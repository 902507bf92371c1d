import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def fdn():
    """The product package; its HIP library must exist (no fallback)."""
    import flowdenoising_amd
    flowdenoising_amd._lib.load()
    return flowdenoising_amd


def rel_err(a, b):
    """max |a-b| / max |b|: the volume-range-relative error the 1e-4 bar is stated in."""
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
