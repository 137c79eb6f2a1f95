import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Vectors recorded from the reference itself (tests/golden/make_golden.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_more():
    """Second batch recorded from the reference (tests/golden/make_golden_more.py): kNN + ratio test, the older
    per-pair API, loop-closure candidate validation, fewer than 6 supporters."""
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors_more.npz"), allow_pickle=False)
