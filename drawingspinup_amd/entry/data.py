"""Host-side data glue of the entry points (PIL + numpy; no torchvision / cv2 needed).

  * SingleImageDataset pieces (mvdiffusion/data/single_image_dataset.py:97-185): RGBA ->
    white-background 256x256 tensor, task one-hots, camera embeddings (elevation_cond,
    delta elevation, delta azimuth) from world->camera poses.
  * DatasetFullImages (3_style_translator/training/data.py:12-51) + overlap_edge_on_img and
    to_image_space (custom_transforms.py:8-9,31-36).
  * the mv -> recon hand-off (instant_nsr/datasets/ortho.py:54-97).
"""
import math
import os

import numpy as np
import torch
from PIL import Image, ImageOps

VIEWS = ["front", "front_right", "right", "back", "left", "front_left"]


# ------------------------------------------------------------------ SingleImageDataset
def load_image_rgba(image_input, size=256, bg=(1.0, 1.0, 1.0)):
    """single_image_dataset.py:97-130 with crop_size == -1: PIL default (bicubic) resize,
    alpha-composite onto the background colour.  Returns (img (H,W,3), alpha (H,W,1)) f32."""
    image_input = image_input.resize((size, size))
    img = np.array(image_input).astype(np.float32) / 255.0
    assert img.shape[-1] == 4, "an RGBA drawing is expected (char/ffc_resnet_inpainted.png)"
    alpha = img[..., 3:4]
    img = img[..., :3] * alpha + np.asarray(bg, np.float32) * (1 - alpha)
    return torch.from_numpy(img), torch.from_numpy(alpha)


def cartesian_to_spherical(xyz):
    xy = xyz[:, 0] ** 2 + xyz[:, 1] ** 2
    z = np.sqrt(xy + xyz[:, 2] ** 2)
    theta = np.arctan2(np.sqrt(xy), xyz[:, 2])
    azimuth = np.arctan2(xyz[:, 1], xyz[:, 0])
    return np.array([theta, azimuth, z])


def get_T(target_RT, cond_RT):
    """single_image_dataset.py:67-80: (delta elevation, delta azimuth mod 2pi)."""
    R, T = target_RT[:3, :3], target_RT[:, -1]
    T_target = -R.T @ T
    R, T = cond_RT[:3, :3], cond_RT[:, -1]
    T_cond = -R.T @ T
    th_c, az_c, _ = cartesian_to_spherical(T_cond[None, :])
    th_t, az_t, _ = cartesian_to_spherical(T_target[None, :])
    return th_t - th_c, (az_t - az_c) % (2 * math.pi)


def mv_batch(single_image, pose_dir=None, size=256):
    """The (12,3,H,W) image batch and (12,5) camera/task embedding of mv.py:70-78.
    pose_dir = .../mvdiffusion/data/fixed_poses/nine_views (the reference's files); without it
    the pipeline's built-in table (pipeline_mvdiffusion_image.py:136-148) is used, which holds
    the same values rounded to f16."""
    from ..mv.pipeline import DEFAULT_CAMERA_EMBEDDING
    img, _ = load_image_rgba(single_image, size)
    imgs = img.permute(2, 0, 1)[None].repeat(12, 1, 1, 1)
    if pose_dir is None:
        return imgs, DEFAULT_CAMERA_EMBEDDING.float()
    poses = {v: np.loadtxt(os.path.join(pose_dir, f"000_{v}_RT.txt")) for v in VIEWS}
    elev, azim = [], []
    for v in VIEWS:
        e, a = get_T(poses[v], poses["front"])
        elev.append(float(e[0])); azim.append(float(a[0]))
    cam = torch.tensor([[0.0, e, a] for e, a in zip(elev, azim)], dtype=torch.float32)
    cam = torch.cat([cam, cam], 0)
    task = torch.cat([torch.tensor([[1.0, 0.0]]).repeat(6, 1), torch.tensor([[0.0, 1.0]]).repeat(6, 1)])
    return imgs, torch.cat([cam, task], -1)


ADD_GRAY_UIDS = ("0b39d3ae37ee430dbe721cdcc40e270c", "b2f0411a69b149088282f262b77970a7",
                 "7d64695e10134f4883cf0f646c21ed30")        # mv.py:59-61


def add_gray(img):
    """mv.py:153-158: darken the colours to 80 % over white, alpha kept."""
    a = np.array(img, dtype=np.float32)
    rgb = a[:, :, 0:3] * 0.8
    mask = a[:, :, 3:4] / 255.0
    a[:, :, 0:3] = rgb * mask + 255 * (1 - mask)
    return Image.fromarray(a.astype(np.uint8))


def fill_holes(fg):
    """fg (H,W) bool tensor -> fg plus every background region that is NOT connected (4-neighbour)
    to the image border, i.e. scipy.ndimage.binary_fill_holes.  Device-friendly flood fill: the
    set of background pixels reachable from the border is grown along whole runs of a row, then of
    a column, until it stops changing (a handful of sweeps for a character silhouette)."""
    bg = ~fg
    H, W = bg.shape
    reach = torch.zeros_like(bg)
    reach[0], reach[-1], reach[:, 0], reach[:, -1] = bg[0], bg[-1], bg[:, 0], bg[:, -1]

    def sweep(r, b):                                     # along dim 1
        n_rows, n = b.shape
        run = torch.cumsum((~b).to(torch.int64), 1)      # run id: increments at every foreground pixel
        key = run + torch.arange(n_rows, device=b.device)[:, None] * (n + 1)
        hit = torch.zeros(n_rows * (n + 1), dtype=torch.bool, device=b.device)
        hit[key[r & b]] = True
        return hit[key] & b

    for _ in range(4 * (H + W)):                         # bound; real shapes need < 10 rounds
        new = sweep(reach, bg)
        new = sweep(new.t().contiguous(), bg.t().contiguous()).t()
        if bool((new == reach).all()):
            break
        reach = new
    return ~reach


def side_mask_from_prediction(image_pil, threshold=12):
    """Foreground matte of a predicted side view.  The reference runs the isnet-dis ONNX matting
    network on the predicted colour image (mv.py:105-150, a third-party CPU model whose weights
    are not in the snapshot); it returns a FILLED silhouette.  The diffusion model renders its
    views on a white background, so the stand-in here is: not-white pixels (any channel more than
    `threshold` grey levels below white) plus everything they enclose — white or paper-coloured
    interiors of a drawing are part of the character, not holes for the mask / opacity losses to
    carve through the body.  `write_mv_outputs(..., matting_fn=...)` takes the real matting model."""
    a = np.array(image_pil.convert("RGB"), np.int16)
    fg = torch.from_numpy((255 - a).max(-1) > threshold)
    return Image.fromarray((fill_holes(fg).numpy() * 255).astype(np.uint8), "L")


def tensor2pil(t):                      # mv.py:47-49
    nd = t.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()
    return Image.fromarray(nd)


NORMAL_MATTE_UIDS = ("01522711d3b642ddbfb506307a007990", "1a2fd47487a24c4c84f2c7d0f7d35147",
                     "1f1654afb5aa4f8daa5db9a96351c226", "d77b86a6b2024cffa36f010e72c0a2af")   # mv.py:118-122


def write_mv_outputs(out_dir, normals, colors, single_image, res=(1024, 1024), uid=None,
                     matting_fn=None):
    """mv.py:105-126 — LANCZOS 1024^2 PNGs; masks: the input alpha (front), mirrored (back), and for
    the four side views a matte of the PREDICTED image (colour, or the normal map for the uids of
    mv.py:118-122): `matting_fn(pil) -> 'L' image` (the reference's isnet-dis session), default
    `side_mask_from_prediction`."""
    matting_fn = matting_fn or side_mask_from_prediction
    for sub in ("normal", "color", "mask"):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    mask_front = single_image.split()[-1]
    mask_back = ImageOps.mirror(mask_front)
    for j, view in enumerate(VIEWS):
        normal = tensor2pil(normals[j].float()).resize(res, Image.LANCZOS)
        color = tensor2pil(colors[j].float()).resize(res, Image.LANCZOS)
        if view == "front":
            m = mask_front.resize(res, Image.NEAREST)
        elif view == "back":
            m = mask_back.resize(res, Image.NEAREST)
        else:
            m = matting_fn(normal if uid in NORMAL_MATTE_UIDS else color)
        normal.save(os.path.join(out_dir, "normal", f"{view}.png"))
        color.save(os.path.join(out_dir, "color", f"{view}.png"))
        m.save(os.path.join(out_dir, "mask", f"{view}.png"))


# ------------------------------------------------------------------ ortho dataset (recon.py)
# instant_nsr/datasets/ortho.py:113-127: drawings reconstructed from a subset of the six views
TWO_VIEW_UIDS = ("025dc91b146d4f57bd114e07165ff7bd", "b03fed9c34f64114a62c7a963fa804e5",
                 "e91d8a6d3aa444f9b10f3a14a6e0a287")
FOUR_VIEW_UIDS = ("b32e37e2f0354f569ea9265d753891f7", "b718c3fb937a416b9fe49ff984a1504e",
                  "d12bed5708ed42f2b615b7911c0291fa", "d2f443e21595431f9f2cd580f291f51b")


def view_types_for(uid):
    if uid in TWO_VIEW_UIDS:
        return ["front", "back"]
    if uid in FOUR_VIEW_UIDS:
        return ["front", "front_right", "back", "front_left"]
    return list(VIEWS)


def load_mv_prediction(mv_dir, device, pose_dir=None, size=(1024, 1024), uid=None):
    """instant_nsr/datasets/ortho.py:54-97,113-127 -> drawingspinup_amd.nsr.system.OrthoData
    (the view subset of the uid's special cases; all view weights 1)."""
    from ..nsr import system as S
    imgs, masks, normals, poses = [], [], [], []

    def pose(v):
        if pose_dir is not None:
            return np.loadtxt(os.path.join(pose_dir, f"000_{v}_RT.txt"))
        return S.ideal_w2c(v)

    front_c2w = S.inv_rt(S.rt_opengl2opencv(pose("front")))[:3, :3]
    for v in view_types_for(uid):
        normal = np.array(Image.open(os.path.join(mv_dir, "normal", f"{v}.png")).convert("RGB"), np.float32)
        normal = normal / 255.0 * 2 - 1
        mask = np.array(Image.open(os.path.join(mv_dir, "mask", f"{v}.png")).convert("L"))
        normal[mask == 0] = 0
        n_cv = normal * np.array([1, -1, -1], np.float32)
        normals.append(torch.from_numpy((n_cv.reshape(-1, 3) @ front_c2w.T).reshape(*normal.shape).astype(np.float32)))
        masks.append(torch.from_numpy(mask > 127))
        img = np.array(Image.open(os.path.join(mv_dir, "color", f"{v}.png")).convert("RGB"), np.float32) / 255.0
        imgs.append(torch.from_numpy(img))
        poses.append(torch.from_numpy(S.inv_rt(S.rt_opengl2opencv(pose(v)))).float())
    return S.OrthoData(torch.stack(imgs), torch.stack(masks), torch.stack(normals), torch.stack(poses), device)


# ------------------------------------------------------------------ DatasetFullImages
def to_image_space(x):                  # custom_transforms.py:8-9
    return ((np.clip(x, -1, 1) + 1) / 2 * 255).astype(np.uint8)


def _to_tensor(pil):                    # torchvision ToTensor
    a = np.array(pil, np.float32) / 255.0
    if a.ndim == 2:
        a = a[..., None]
    return torch.from_numpy(a).permute(2, 0, 1)


def _rgb_normalised(pil):               # rgba_to_rgb + ToTensor + Normalize(0.5, 0.5)
    if pil.mode == "RGBA":
        pil = Image.fromarray(np.array(pil)[..., :3])
    return (_to_tensor(pil) - 0.5) / 0.5


def overlap_edge_on_img(edge, img):     # custom_transforms.py:31-36
    edge_mask = np.array(edge) < 255
    img = np.array(img)
    img[edge_mask, 0:3] = 0
    img[edge_mask, 3] = 255
    return Image.fromarray(img)


class DatasetFullImages:
    """training/data.py:12-51."""

    def __init__(self, data_root, pre_dir, use_mask=False, use_pos=False, use_edge=False):
        self.data_root, self.pre_dir = data_root, pre_dir
        self.use_mask, self.use_pos, self.use_edge = use_mask, use_pos, use_edge
        self.fnames = sorted(os.listdir(os.path.join(data_root, "color")))

    def __len__(self):
        return len(self.fnames)

    def __getitem__(self, item):
        name = self.fnames[item]
        pre_color = Image.open(os.path.join(self.data_root, self.pre_dir, name))
        mask = pre_color.split()[-1]
        if self.use_edge:
            pre_color = overlap_edge_on_img(Image.open(os.path.join(self.data_root, "edge", name)), pre_color)
        feats = [_rgb_normalised(pre_color)]
        if self.use_mask:
            feats.append(_to_tensor(mask))
        if self.use_pos:
            feats.append(_rgb_normalised(Image.open(os.path.join(self.data_root, "pos", name)))[0:2])
        return {"file_name": name, "pre": torch.cat(feats, 0), "pre_mask": _to_tensor(mask)}
