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
    # The driver runs `-m gpu -x`: the parity tests of the SURVEY.md 8(a) rows (HIP vs oracle /
    # reference fixture) come first, self-comparisons (native vs Python-sequenced) after them,
    # statistical / convergence properties last, so that a late failure cannot hide parity evidence.
    def rank(item):
        f = item.nodeid.split("::")[0].rsplit("/", 1)[-1]
        name = item.nodeid.lower()
        late = any(k in name for k in ("moments", "converges", "timing_counters", "statistic"))
        return (2 if late else 1 if f in _SELF_COMPARISON_FILES else 0,
                _PARITY_ORDER.index(f) if f in _PARITY_ORDER else len(_PARITY_ORDER))
    items.sort(key=rank)      # stable: the order inside a file is kept


_PARITY_ORDER = [
    "test_gpu_hashgrid.py", "test_gpu_nsr_reference_step.py", "test_gpu_render.py",
    "test_gpu_nsr_step.py", "test_gpu_nsr_model.py", "test_gpu_mesh.py", "test_gpu_unet.py",
    "test_gpu_attention.py", "test_gpu_conv_f16.py", "test_gpu_style.py", "test_gpu_shims.py", "test_gpu_matting.py",
    "test_gpu_thinning.py", "test_gpu_mesh_post.py", "test_gpu_decimate.py",
    "test_gpu_style_train.py", "test_contour_host.py", "test_contour_inpaint.py",
]
_SELF_COMPARISON_FILES = {"test_gpu_nsr_native.py", "test_gpu_entry.py"}


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
