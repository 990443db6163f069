"""Host glue of the contour-removal stage (1_lama_contour_remover/predict.py:35-67 and
saicinpainting/training/data/datasets.py:44-74, aug.py:78-105).

    texture.png (RGBA) --prepare_input--> (1,4,512,512) --FFC-ResNet--> contour probability
      --contour_masks--> predicted contour mask, inpaint mask = max(contour, 255 - alpha)

      --inpaint--> the drawing without its contour lines (RGBA, predict.py:63-66)

The last step of the reference, `cv2.inpaint(img, inpaint_mask, 3, cv2.INPAINT_TELEA)`
(predict.py:63), is OpenCV's CPU fast-marching inpainting.  `inpaint` runs the library's own
restatement of it (csrc/inpaint_telea.hip, host code behind the C ABI as in the reference; OpenCV
is not installed here, so it is unpinned against the real cv2 — oracle/telea_ref.py documents what
the tests do check).
"""
import ctypes as C

import numpy as np
import torch
from PIL import Image

from .ffc import LAMA_FOURIER_GENERATOR, make_generator


def prepare_input(rgba_image, size=512):
    """InpaintingDrawingsDataset.__getitem__: the drawing pasted on white through its own alpha,
    alpha as the mask channel, both resized to size x size (bicubic) and scaled to [0,1]
    (transforms.Grayscale(1) on an 'L' alpha image is the identity)."""
    img = rgba_image if isinstance(rgba_image, Image.Image) else Image.fromarray(rgba_image)
    if img.mode != "RGBA":
        raise ValueError("the drawing must carry its mask as the alpha channel (RGBA)")
    rgb = Image.new("RGB", img.size, (255, 255, 255))
    rgb.paste(img, (0, 0), img)
    mask = img.split()[-1]
    rgb = rgb.resize((size, size), Image.BICUBIC)
    mask = mask.resize((size, size), Image.BICUBIC)
    x = np.concatenate([np.asarray(rgb, np.float32) / 255.0,
                        np.asarray(mask, np.float32)[..., None] / 255.0], -1)
    return torch.from_numpy(x).permute(2, 0, 1)[None].contiguous()


@torch.no_grad()
def contour_masks(model, inp, threshold=0.2):
    """predict.py:51-62: contour = (prob > 0.2) * 255, inpaint mask = max(contour, 255 - alpha).
    Returns (img uint8 (H,W,3), alpha uint8 (H,W), contour uint8 (H,W), inpaint_mask uint8 (H,W))
    on the host, as predict.py builds them."""
    dev = next(model.parameters()).device
    prob = model(inp.to(dev))[0, 0].float().cpu().numpy()
    x = inp[0].permute(1, 2, 0).cpu().numpy()
    img = (x[:, :, 0:3] * 255).astype("uint8")
    alpha = (x[:, :, 3] * 255).astype("uint8")
    contour = np.clip((prob > threshold) * 255, 0, 255).astype("uint8")
    return img, alpha, contour, np.maximum(contour, 255 - alpha).astype(np.uint8)


def inpaint(img, inpaint_mask, radius=3):
    """cv2.inpaint(img, inpaint_mask, radius, cv2.INPAINT_TELEA): img (H,W,3) uint8, mask (H,W) uint8
    (non-zero = fill) on the host -> (H,W,3) uint8."""
    from .._lib import check, lib
    img = np.ascontiguousarray(img, np.uint8)
    mask = np.ascontiguousarray(inpaint_mask, np.uint8)
    if img.ndim != 3 or img.shape[2] != 3 or mask.shape != img.shape[:2]:
        raise ValueError("inpaint: img (H,W,3) uint8 and mask (H,W) uint8 expected")
    out = np.empty_like(img)
    check(lib().dsu_inpaint_telea_u8c3(img.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p),
                                       img.shape[0], img.shape[1], int(radius),
                                       out.ctypes.data_as(C.c_void_p)), "dsu_inpaint_telea_u8c3")
    return out


def remove_contour(model, rgba_image, size=512, threshold=0.2, radius=3):
    """predict.py:47-66 for one drawing: RGBA uint8 (size, size, 4) with the contour lines (and the
    background) inpainted from the character's own pixels; alpha = the resized input alpha."""
    img, alpha, _, mask = contour_masks(model, prepare_input(rgba_image, size), threshold)
    return np.concatenate([inpaint(img, mask, radius), alpha[..., None]], 2)


def load_generator(checkpoint=None, device="cuda", config=None):
    """predict.py:14-18 `load_checkpoint` (strict=False there; the shipped checkpoint holds exactly
    this module tree, so a strict load is what is tested)."""
    model = make_generator(**(config or LAMA_FOURIER_GENERATOR))
    if checkpoint is not None:
        model.load_state_dict(torch.load(checkpoint, map_location="cpu"), strict=False)
    return model.eval().to(device)
