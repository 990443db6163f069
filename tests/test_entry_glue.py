"""CPU: host-side data glue of the entry points (no GPU, synthetic files)."""
import os

import numpy as np
import torch
from PIL import Image

from drawingspinup_amd.entry import data as D


def _rgba(size=64):
    g = np.random.default_rng(0)
    a = np.zeros((size, size, 4), np.uint8)
    a[..., :3] = g.integers(0, 256, (size, size, 3))
    yy, xx = np.mgrid[:size, :size]
    a[..., 3] = (((xx - size / 2) ** 2 + (yy - size / 2) ** 2) < (size / 3) ** 2) * 255
    return Image.fromarray(a, "RGBA")


def test_mv_batch_and_camera_embedding():
    img = _rgba(96)
    imgs, cam = D.mv_batch(img, None, 32)
    assert imgs.shape == (12, 3, 32, 32) and cam.shape == (12, 5)
    assert torch.equal(imgs[0], imgs[11])                       # 12 identical conditioning images
    assert float(imgs[0][:, 0, 0].min()) == 1.0                  # transparent corner -> white
    # pose files -> same table as the pipeline default (to f16 rounding): SURVEY.md §8c KAT
    ref = "/root/reference/2_charactor_reconstructor/mvdiffusion/data/fixed_poses/nine_views"
    if os.path.isdir(ref):
        _, cam2 = D.mv_batch(img, ref, 32)
        np.testing.assert_allclose(cam2.numpy(), cam.numpy(), atol=2.5e-3)
        np.testing.assert_allclose(cam2[1, 1:3].numpy(), [-0.23624, 0.81238], atol=1e-4)
        np.testing.assert_allclose(cam2[4, 1:3].numpy(), [0.69066, 4.83508], atol=1e-4)


def test_dataset_full_images_and_mv_handoff(tmp_path):
    root = tmp_path / "act"
    for sub in ("color", "pos", "edge"):
        os.makedirs(root / sub)
    img = _rgba(32)
    img.save(root / "color" / "0001.png")
    _rgba(32).save(root / "pos" / "0001.png")
    edge = np.full((32, 32), 255, np.uint8)
    edge[5, :] = 0
    Image.fromarray(edge).save(root / "edge" / "0001.png")
    b1 = D.DatasetFullImages(str(root), "color", True, True, False)[0]
    assert b1["pre"].shape == (6, 32, 32) and b1["pre_mask"].shape == (1, 32, 32)
    assert float(b1["pre"][:3].min()) >= -1 and float(b1["pre"][:3].max()) <= 1
    assert set(np.unique(b1["pre"][3].numpy())) <= {0.0, 1.0}
    b2 = D.DatasetFullImages(str(root), "color", True, True, True)[0]
    assert torch.equal(b2["pre"][:3, 5, :], -torch.ones(3, 32))   # edge pixels painted black
    assert np.array_equal(D.to_image_space(np.array([-2.0, -1.0, 0.0, 1.0, 3.0])), [0, 0, 127, 255, 255])
    # mv outputs -> ortho dataset
    out = tmp_path / "mv"
    n = torch.rand(6, 3, 16, 16)
    c = torch.rand(6, 3, 16, 16)
    D.write_mv_outputs(str(out), n, c, img, res=(64, 64))
    assert sorted(os.listdir(out / "color")) == sorted(f"{v}.png" for v in D.VIEWS)
    ds = D.load_mv_prediction(str(out), "cpu")
    assert ds.all_images.shape == (6, 64, 64, 3) and ds.all_masks.shape == (6, 64, 64)
    assert ds.all_c2w.shape == (6, 3, 4) and ds.all_normals_world.shape == (6, 64, 64, 3)
    m = ds.all_masks[0] > 0
    assert float(ds.all_normals_world[0][~m].abs().max()) == 0.0   # normals zeroed outside the mask
