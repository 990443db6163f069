"""Fixture for the export's colour back-projection and thinning offsets: runs the
REFERENCE's own `color_projection` (2_charactor_reconstructor/instant_nsr/utils/coloring_utils.py:
91-138, with load_color / get_color_from_image / direct_query / interpolate_rgb) and
`get_offset_mask` (instant_nsr/utils/thinning_utils.py:96-193) on a small seeded mesh with
self-occlusion along z.

    python tests/golden/make_mesh_color_golden.py     # needs /root/reference; ~1 min

Real libraries used as the reference uses them: PIL (LANCZOS resizes), scipy (cKDTree), numpy.
Absent third-party pieces are served by oracle/mesh_post_ref.py (mesh_raycast, the pytorch3d
silhouette renderer, cv2's elliptic erosion / flip): parity unpinned for those ops alone.  The
fixture stores the inputs (mesh, the three PNG-equivalent images, skeleton samples) and the
reference's outputs (per-vertex colours; offset values and mask for the three thinning types).
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
UTILS = "/root/reference/2_charactor_reconstructor/instant_nsr/utils"
from oracle import mesh_post_ref as R  # noqa: E402
RES = 512          # image side of the fixture (the reference hard-codes 2048: patched below)


def stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def fixture_mesh():
    """A body (ellipsoid) with an arm in front of it (second blob at larger z, overlapping in xy):
    vertices of the body behind the arm are hidden from +z, so all three colouring branches run."""
    import torch
    from drawingspinup_amd.nsr import mesh as M
    n = 40
    c = torch.linspace(-0.5, 0.5, n, dtype=torch.float64)
    x, y, z = torch.meshgrid(c, c, c, indexing="ij")
    body = ((x / 0.30) ** 2 + (y / 0.40) ** 2 + (z / 0.12) ** 2).sqrt() - 1.0
    arm = (((x - 0.12) / 0.10) ** 2 + ((y + 0.05) / 0.22) ** 2 + ((z - 0.27) / 0.07) ** 2).sqrt() - 1.0
    vol = -torch.minimum(body * 0.12, arm * 0.07)                # positive inside
    v, f = M.marching_cubes(vol, 0.0)
    v = v / (n - 1) - 0.5
    v[:, 2] *= 0.2                     # a flat character: 0.05 thick, inside the thinning window (< 0.06)
    # marching-cubes vertices sit on lattice lines: front and back vertices share their (x, y) bit
    # for bit and the 8-nearest-neighbour sets of interpolate_rgb are full of exact distance ties
    # (cKDTree's choice among them is arbitrary).  save_mesh smooths the mesh before colouring, so
    # real inputs are in general position: a small seeded displacement does the same here.
    v = v.numpy()
    v[:, :2] += np.random.default_rng(2).normal(0.0, 3e-4, (len(v), 2))
    return v, f.numpy()


def fixture_images(seed=0):
    rng = np.random.default_rng(seed)
    def smooth(shape):
        a = rng.random((16, 16, 3))
        img = Image.fromarray((a * 255).astype(np.uint8)).resize(shape, Image.BICUBIC)
        return np.array(img)
    yy, xx = np.mgrid[0:256, 0:256]
    mask = ((((xx - 127.5) / 85.0) ** 2 + ((yy - 127.5) / 112.0) ** 2) <= 1.0).astype(np.uint8) * 255
    return smooth((256, 256)), smooth((256, 256)), mask


def main():
    stub("cv2", MORPH_ELLIPSE=R.MORPH_ELLIPSE, getStructuringElement=R.getStructuringElement,
         erode=R.erode, dilate=R.dilate, flip=R.flip)
    stub("trimesh")
    stub("mesh_raycast", raycast=R.raycast)
    stub("igl")
    stub("skimage", morphology=types.ModuleType("skimage.morphology"))
    stub("skimage.morphology")
    stub("pytorch3d")
    stub("pytorch3d.structures", Meshes=object)
    stub("pytorch3d.renderer", RasterizationSettings=lambda **k: None, MeshRasterizer=lambda **k: None)
    stub("pytorch3d.renderer.cameras", look_at_view_transform=lambda *a: (None, None),
         OrthographicCameras=lambda **k: None)

    def load(name):
        spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(UTILS, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    cu, tu = load("coloring_utils"), load("thinning_utils")
    # the reference hard-codes 2048^2 images: same code at the fixture's resolution
    cu.mask_renderer = R.MaskRenderer(RES)
    _open = Image.open

    class _Resized:
        def __init__(self, img):
            self.img = img

        def resize(self, size, resample):
            assert size == (2048, 2048) and resample == Image.LANCZOS
            return self.img.resize((RES, RES), resample)
    cu.Image = types.SimpleNamespace(open=lambda p: _Resized(_open(p)), LANCZOS=Image.LANCZOS)

    verts, faces = fixture_mesh()
    cf, cb, mk = fixture_images()
    out = {"verts": verts, "faces": faces, "color_front": cf, "color_back": cb, "mask_front": mk,
           "res": np.int64(RES)}
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "color")); os.makedirs(os.path.join(d, "mask"))
        Image.fromarray(cf).save(os.path.join(d, "color", "front.png"))
        Image.fromarray(cb).save(os.path.join(d, "color", "back.png"))
        Image.fromarray(mk, "L").save(os.path.join(d, "mask", "front.png"))
        colors = cu.color_projection(verts.copy(), faces, d)
        # (the product takes the three LANCZOS-resized images as inputs; the test repeats the
        # resizes with PIL from the small images stored here)
    out["vert_colors"] = colors
    # thinning offsets: skeleton samples along the body's vertical axis and across the arm
    rng = np.random.default_rng(1)
    thin = np.concatenate([np.stack([np.zeros(40), np.linspace(-0.35, 0.35, 40)], 1),
                           np.stack([np.linspace(0.05, 0.2, 20), np.full(20, -0.05)], 1),
                           rng.uniform(-0.45, 0.45, (30, 2))], 0)
    dists = rng.uniform(0.001, 0.012, len(thin))
    out["thin_coords"], out["coord_dists"], out["min_thickness"] = thin, dists, np.float64(1.0 / 512)
    for ty in ("double", "front", "back"):
        ov, om = tu.get_offset_mask(verts.copy(), faces, thin, dists, 1.0 / 512, ty)
        out["offset_values_" + ty], out["offset_mask_" + ty] = ov, om
    np.savez_compressed(os.path.join(HERE, "mesh_color_reference.npz"), **out)
    known = np.count_nonzero(np.abs(colors).sum(1))
    print("wrote mesh_color_reference.npz:", verts.shape, faces.shape,
          {k: int(out["offset_mask_" + k].sum()) for k in ("double", "front", "back")}, known)


if __name__ == "__main__":
    main()
