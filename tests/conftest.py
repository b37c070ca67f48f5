import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_golden.npz"))


@pytest.fixture(scope="session")
def ctx():
    """HIP context; the product path has no CPU fallback, so this fails loudly without a GPU."""
    from tda_eeg_audio_amd import _lib
    return _lib.get_ctx(0)
