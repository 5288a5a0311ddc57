import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The C-ABI library (built in-tree by __graft_entry__.build(); built here if the snapshot lacks it --
    abi.load() itself never builds or falls back, it raises when the library is missing)."""
    from vae_captioning_amd import abi
    if not os.path.exists(abi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return abi.load()


@pytest.fixture(autouse=True)
def _release_device_temporaries():
    yield
    try:
        from tests import gpu_util
    except Exception:
        return
    gpu_util.release()
