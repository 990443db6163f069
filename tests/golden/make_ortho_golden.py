"""Fixture for the NSR dataset loader: runs the REFERENCE's own `load_a_prediction`
(2_charactor_reconstructor/instant_nsr/datasets/ortho.py:54-97, with its helpers
RT_opengl2opencv / inv_RT / normal_opengl2opencv / camNormal2worldNormal and the fixed camera
poses of instant_nsr/datasets/fixed_poses) on a small seeded set of mv outputs.  cv2 /
pytorch_lightning / the dataset registry are imported by that file but not used by the function:
stubbed.

    python tests/golden/make_ortho_golden.py      # needs /root/reference; writes ortho_reference.npz
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
from PIL import Image

REF = "/root/reference/2_charactor_reconstructor"
HERE = os.path.dirname(os.path.abspath(__file__))
VIEWS = ["front", "front_right", "right", "back", "left", "front_left"]


def write_seeded_mv(out_dir, size=24, seed=5):
    """colour / normal / mask PNGs as mv.py writes them (RGB, RGB, L)."""
    rng = np.random.default_rng(seed)
    for sub in ("color", "normal", "mask"):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    for v in VIEWS:
        Image.fromarray(rng.integers(0, 256, (size, size, 3), dtype=np.uint8)).save(os.path.join(out_dir, "color", v + ".png"))
        Image.fromarray(rng.integers(0, 256, (size, size, 3), dtype=np.uint8)).save(os.path.join(out_dir, "normal", v + ".png"))
        m = rng.integers(0, 256, (size, size), dtype=np.uint8)
        m[:4] = 0
        m[-4:] = 255
        Image.fromarray(m).save(os.path.join(out_dir, "mask", v + ".png"))


def main():
    for name in ("cv2", "pytorch_lightning"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["pytorch_lightning"].LightningDataModule = object
    sys.path.insert(0, REF)
    pkg = types.ModuleType("instant_nsr")
    pkg.__path__ = [os.path.join(REF, "instant_nsr")]
    sys.modules["instant_nsr"] = pkg
    ds = types.ModuleType("instant_nsr.datasets")
    ds.register = lambda name: (lambda cls: cls)
    sys.modules["instant_nsr.datasets"] = ds
    misc = types.ModuleType("instant_nsr.utils.misc")
    misc.get_rank = lambda: 0
    sys.modules["instant_nsr.utils"] = types.ModuleType("instant_nsr.utils")
    sys.modules["instant_nsr.utils.misc"] = misc
    models = types.ModuleType("instant_nsr.models")
    models.__path__ = [os.path.join(REF, "instant_nsr", "models")]
    sys.modules["instant_nsr.models"] = models
    spec = importlib.util.spec_from_file_location("ref_ortho", os.path.join(REF, "instant_nsr", "datasets", "ortho.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    pose_dir = os.path.join(REF, "instant_nsr", "datasets", "fixed_poses")
    poses = {v: np.loadtxt(os.path.join(pose_dir, f"000_{v}_RT.txt")) for v in VIEWS}
    with tempfile.TemporaryDirectory() as tmp:
        write_seeded_mv(tmp)
        out = ref.load_a_prediction(tmp, [24, 24], VIEWS, pose_dir)
    names = ("images", "masks", "normals_cam", "normals_world", "poses", "w2cs", "origins", "directions")
    np.savez_compressed(os.path.join(HERE, "ortho_reference.npz"),
                        **{n: np.asarray(a) for n, a in zip(names, out)},
                        **{"RT_" + v: p for v, p in poses.items()})
    print("wrote ortho_reference.npz", [np.asarray(a).shape for a in out])


if __name__ == "__main__":
    main()
