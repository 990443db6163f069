import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def pytest_collection_modifyitems(config, items):
    """A test that asks for the device fixture but lacks the gpu mark would be deselected by
    `-m gpu` and skipped by `-m "not gpu"`: it would never run anywhere.  Refuse to collect it."""
    bad = [it.nodeid for it in items
           if "dev" in getattr(it, "fixturenames", ()) and it.get_closest_marker("gpu") is None]
    if bad:
        raise pytest.UsageError("tests use the `dev` fixture without @pytest.mark.gpu: " + ", ".join(bad))


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
