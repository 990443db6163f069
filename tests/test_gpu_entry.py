"""The reference's entry points (mv.py -> recon.py -> test_stage1.py -> test_stage2.py) on a
synthetic uid directory: same CLI flags, same on-disk layout, files handed over through PNGs."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def _drawing(size=512):
    g = np.random.default_rng(0)
    a = np.zeros((size, size, 4), np.uint8)
    a[..., :3] = np.kron(g.integers(0, 256, (size // 8, size // 8, 3)), np.ones((8, 8, 1))).astype(np.uint8)
    yy, xx = np.mgrid[:size, :size]
    a[..., 3] = ((((xx - size / 2) / (0.3 * size)) ** 2 + ((yy - size / 2) / (0.42 * size)) ** 2) <= 1) * 255
    return Image.fromarray(a, "RGBA")


def test_entry_points_chain(dev, tmp_path):
    from drawingspinup_amd.entry import mv, recon, _test_stage
    root, uid = str(tmp_path), "uid0"
    os.makedirs(os.path.join(root, uid, "char"))
    _drawing().save(os.path.join(root, uid, "char", "ffc_resnet_inpainted.png"))
    mv.main(["--uid", uid, "--data_root", root, "--num_inference_steps", "2", "--random_init"])
    for sub in ("color", "normal", "mask"):
        files = sorted(os.listdir(os.path.join(root, uid, "mv", sub)))
        assert files == sorted(f"{v}.png" for v in ("front", "front_right", "right", "back", "left", "front_left"))
    assert Image.open(os.path.join(root, uid, "mv", "color", "front.png")).size == (1024, 1024)
    # the side masks through the IS-Net session (mv.py:17-18,120-122) instead of the silhouette stand-in
    mv.main(["--uid", uid, "--data_root", root, "--num_inference_steps", "2", "--random_init",
             "--matting", "isnet", "--save_folder", "mv_isnet"])
    for v in ("front_right", "right", "left", "front_left"):
        m = Image.open(os.path.join(root, uid, "mv_isnet", "mask", f"{v}.png"))
        assert m.mode == "L" and m.size == (1024, 1024)
    for v in ("front", "back"):                                       # the input's alpha, as before
        assert np.array_equal(np.array(Image.open(os.path.join(root, uid, "mv_isnet", "mask", f"{v}.png"))),
                              np.array(Image.open(os.path.join(root, uid, "mv", "mask", f"{v}.png"))))
    _drawing().split()[-1].save(os.path.join(root, uid, "char", "mask.png"))     # front mask (ortho.py:153-156)
    import json
    thin_list = os.path.join(root, "drawings_uids_thinning.json")
    json.dump(["some-other-uid"], open(thin_list, "w"))
    # the reference's CLI (--config / --uid) + the two path overrides a synthetic directory needs;
    # every export switch comes from the YAML: remeshing to 50000 faces, smoothing, shearing and
    # colour back-projection on, thinning only for the uids of the thinning list (recon.py:53-65)
    recon.main(["--config", "./configs/neuralangelo-ortho-wmask.yaml", "--uid", uid, "--data_root", root,
                "--thinning_uid_list_file", thin_list, "--max_steps", "20"])
    # recon.py's product: the mesh file, named as neus_ortho.py:184-196 names it
    obj = open(os.path.join(root, uid, "mesh", "it20-mc512-f50000_c_r_s_cbp.obj")).read().splitlines()
    nv, nf = sum(l.startswith("v ") for l in obj), sum(l.startswith("f ") for l in obj)
    assert nv > 1000 and 2000 < nf <= 50000                          # model.geometry.face_count
    # a uid of the thinning list gets the `_t` step and suffix
    json.dump([uid], open(thin_list, "w"))
    recon.main(["--uid", uid, "--data_root", root, "--thinning_uid_list_file", thin_list,
                "--max_steps", "4", "--resolution", "256"])
    assert os.path.isfile(os.path.join(root, uid, "mesh", "it4-mc256-f50000_c_r_t_s_cbp.obj"))
    assert len(obj[0].split()) == 7                                   # v x y z r g b
    idx = np.array([[int(t) for t in l.split()[1:]] for l in obj if l.startswith("f ")])
    assert idx.min() == 1 and idx.max() == nv                         # 1-based, every vertex used
    sd = torch.load(os.path.join(root, uid, "mesh", "it20.ckpt"), map_location="cpu")
    assert "geometry.encoding.encoding.encoding.params" in sd and "variance.variance" in sd
    # stage 3 inputs: 2 synthetic frames (colour / pos / edge) in the blender_render layout
    act = os.path.join(root, uid, "mesh", "blender_render", "dance")
    for sub in ("color", "pos", "edge"):
        os.makedirs(os.path.join(act, sub))
    for f in range(2):
        _drawing().save(os.path.join(act, "color", f"{f:04d}.png"))
        _drawing().save(os.path.join(act, "pos", f"{f:04d}.png"))
        Image.fromarray(np.full((512, 512), 255, np.uint8)).save(os.path.join(act, "edge", f"{f:04d}.png"))
    _test_stage.run(1, ["--uid", uid, "--root_dir", root, "--random_init"])
    s1 = Image.open(os.path.join(act, "res_stage1_mask_pos", "0000.png"))
    assert s1.mode == "RGBA" and s1.size == (512, 512)
    _test_stage.run(2, ["--uid", uid, "--root_dir", root, "--random_init"])
    s2 = Image.open(os.path.join(act, "res_stage2_mask_pos_edge", "0001.png"))
    assert s2.mode == "RGBA" and s2.size == (512, 512)
    assert np.array_equal(np.array(s2)[..., 3], np.array(_drawing())[..., 3])      # alpha = input mask
