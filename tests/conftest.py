import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fa():
    import _pkg
    return _pkg.load()


@pytest.fixture(scope="session")
def po():
    """CPU oracle (checker only)."""
    import _pkg
    return _pkg.load_oracle()


@pytest.fixture(scope="session")
def gpu_lib(fa):
    """libflowagg on a GPU box; fails loudly (no fallback) when the extension is missing."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    if not os.path.exists(fa.LIB_PATH):
        fa.build()
    return fa.lib()
