"""Generate tests/golden/matting_reference.npz by running the REFERENCE's own `remove_background`
and `add_gray` (2_charactor_reconstructor/mv.py:134-158) in this container.

    python tests/golden/make_matting_golden.py          # needs /root/reference

mv.py is imported unmodified.  Its module-level imports that are absent here are stand-ins that
the two functions never touch (omegaconf, diffusers: oracle/stubs) or that ARE the seam under test:
`onnxruntime.InferenceSession` is a stub session whose `run` applies a fixed closed-form map to the
array it is fed (values on both sides of [0, 1], so the clip is exercised) and records that array.
The fixture holds the array the reference fed to the session, the matte it returned and the
add_gray result; the input images are rebuilt from seeds by `synthetic_rgb` / `synthetic_rgba`
below (the CPU test imports them from here).
"""
import os
import sys
import types

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/2_charactor_reconstructor"


def synthetic_rgb(seed=0, w=48, h=40):
    rng = np.random.default_rng(seed)
    return Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB")


def synthetic_rgba(seed=1, w=24, h=20):
    rng = np.random.default_rng(seed)
    return Image.fromarray(rng.integers(0, 256, (h, w, 4), dtype=np.uint8), "RGBA")


def stub_network(x):
    """(1, 3, H, W) float32 in [-0.5, 0.5] -> [(1, 1, H, W)]: some values below 0 and above 1."""
    return [(0.5 + 1.6 * x[:, 0:1] - 0.7 * x[:, 1:2] * x[:, 2:3]).astype(np.float32)]


class StubSession:
    class _In:
        name = "input_image"

    def __init__(self, *a, **k):
        self.fed = None

    def get_inputs(self):
        return [self._In()]

    def run(self, names, feed):
        assert names is None and list(feed) == ["input_image"]
        self.fed = feed["input_image"]
        return stub_network(self.fed)


if __name__ == "__main__":
    ort = types.ModuleType("onnxruntime")
    ort.InferenceSession = StubSession
    sys.modules["onnxruntime"] = ort
    oc = types.ModuleType("omegaconf")
    oc.OmegaConf = type("OmegaConf", (), {})
    sys.modules["omegaconf"] = oc
    sys.path.insert(0, os.path.join(ROOT, "oracle", "stubs"))
    import diffusers  # noqa: F401  (stub package)
    if not hasattr(diffusers, "DiffusionPipeline"):
        diffusers.DiffusionPipeline = type("DiffusionPipeline", (), {})
    cwd = os.getcwd()
    os.chdir(REF)                              # mv.py's dataset import reads relative pose paths
    sys.path.insert(0, REF)
    sys.argv = [sys.argv[0]]
    import mv as ref_mv
    os.chdir(cwd)
    sess = StubSession()
    matte = ref_mv.remove_background(sess, synthetic_rgb())
    gray = ref_mv.add_gray(synthetic_rgba())
    out = {"fed": sess.fed, "matte": np.array(matte), "matte_mode": np.array(matte.mode),
           "add_gray": np.array(gray), "add_gray_mode": np.array(gray.mode)}
    path = os.path.join(ROOT, "tests", "golden", "matting_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape, str(v.dtype)) for k, v in out.items()})
