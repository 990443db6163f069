"""Generate tests/golden/glue_reference.npz by running the REFERENCE's own host glue classes on
synthetic images in this container:

  * SingleImageDataset (2_charactor_reconstructor/mvdiffusion/data/single_image_dataset.py:17-185)
    on a seeded 96x80 RGBA drawing -> imgs_in / alphas / camera and task embeddings (M8);
  * DatasetFullImages + overlap_edge_on_img (3_style_translator/training/data.py:12-51,
    custom_transforms.py:31-36) on three seeded 40x32 colour / pos / edge frames, stage-1 and
    stage-2 flag sets (S6).

    python tests/golden/make_glue_golden.py        # needs /root/reference

The inputs are rebuilt from seeds by `synthetic_drawing_rgba` / `synthetic_frame_set` below (the
CPU test imports them from here), so the fixture holds only the reference's OUTPUTS plus the six
nine_views pose matrices (configuration constants).  torchvision.transforms is shimmed as in
make_style_train_golden.py (ToTensor / Normalize / Compose, published semantics); PIL does the
resizing in both the reference and the product.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_MV = "/root/reference/2_charactor_reconstructor"
REF_ST = "/root/reference/3_style_translator"
VIEWS = ["front", "front_right", "right", "back", "left", "front_left"]


def synthetic_drawing_rgba(seed=0, w=96, h=80):
    rng = np.random.default_rng(seed)
    rgb = rng.integers(0, 256, (h // 8, w // 8, 3), dtype=np.uint8).repeat(8, 0).repeat(8, 1)
    yy, xx = np.mgrid[0:h, 0:w]
    d = ((xx - w / 2) / (0.4 * w)) ** 2 + ((yy - h / 2) / (0.45 * h)) ** 2
    alpha = np.clip((1.1 - d) * 400, 0, 255).astype(np.uint8)            # soft edge: partial alpha
    return Image.fromarray(np.dstack([rgb, alpha]), "RGBA")


def synthetic_frame_set(root, seed=1, n=3, w=40, h=32):
    """<root>/{color,pos,edge,res}/000N.png as run_render.py / test_stage1.py leave them."""
    rng = np.random.default_rng(seed)
    for sub in ("color", "pos", "edge", "res"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    for i in range(n):
        a = (rng.random((h, w)) > 0.3).astype(np.uint8) * 255
        for sub in ("color", "pos", "res"):
            img = np.dstack([rng.integers(0, 256, (h, w, 3), dtype=np.uint8), a])
            Image.fromarray(img, "RGBA").save(os.path.join(root, sub, f"{i:04d}.png"))
        edge = np.where(rng.random((h, w)) > 0.85, rng.integers(0, 255, (h, w)), 255).astype(np.uint8)
        Image.fromarray(edge, "L").save(os.path.join(root, "edge", f"{i:04d}.png"))


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    out = {}
    # ---------------- M8
    cwd = os.getcwd()
    os.chdir(REF_MV)                         # the class reads ./mvdiffusion/data/fixed_poses/...
    sys.path.insert(0, REF_MV)
    from mvdiffusion.data.single_image_dataset import SingleImageDataset
    ds = SingleImageDataset(num_views=6, img_wh=(64, 64), bg_color="white", crop_size=-1,
                            single_image=synthetic_drawing_rgba())
    item = ds[0]
    for k, v in item.items():
        out["mv." + k] = v.numpy()
    for v in VIEWS:
        out["pose." + v] = np.loadtxt(os.path.join(ds.fix_cam_pose_dir, f"000_{v}_RT.txt"))
    os.chdir(cwd)
    # ---------------- S6
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")

    class _Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class _ToTensor:                                   # PIL uint8 HWC -> float CHW / 255
        def __call__(self, pic):
            a = np.array(pic, np.float32) / 255.0
            if a.ndim == 2:
                a = a[..., None]
            return torch.from_numpy(a).permute(2, 0, 1).contiguous()

    class _Normalize:
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean).view(-1, 1, 1)
            self.std = torch.tensor(std).view(-1, 1, 1)

        def __call__(self, t):
            return (t - self.mean) / self.std

    tv.transforms.Compose, tv.transforms.ToTensor, tv.transforms.Normalize = _Compose, _ToTensor, _Normalize
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tv.transforms
    sys.modules["cv2"] = types.ModuleType("cv2")        # imported by custom_transforms, unused here
    sys.path.insert(0, REF_ST)
    from training import data as ref_data
    with tempfile.TemporaryDirectory() as root:
        synthetic_frame_set(root)
        for stage, (pre, flags) in {"stage1": ("color", dict(use_mask=True, use_pos=True, use_edge=False)),
                                    "stage2": ("res", dict(use_mask=True, use_pos=True, use_edge=True))}.items():
            dsf = ref_data.DatasetFullImages(root, pre, **flags)
            out[stage + ".len"] = np.int64(len(dsf))
            for i in range(len(dsf)):
                it = dsf[i]
                out[f"{stage}.{i}.pre"] = it["pre"].numpy()
                out[f"{stage}.{i}.pre_mask"] = it["pre_mask"].numpy()
                out[f"{stage}.{i}.file_name"] = np.array(it["file_name"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "glue_reference.npz"), **out)
    print("wrote glue_reference.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})
