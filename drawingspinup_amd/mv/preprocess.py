"""Image preparation of MVDiffusionImagePipeline._encode_image (pipeline_mvdiffusion_image.py:
150-182) as tensor programs (host or device, integer arithmetic where the reference's is).

The reference hands the pipeline a float16 tensor (mv.py:70,79), turns every row into a PIL image
with torchvision's `to_pil_image` (`pic.mul(255).byte()`: the product is formed in the tensor's own
dtype and TRUNCATED, pipeline :358), and from those 8-bit images takes
  * the CLIP input: `CLIPImageProcessor` = Pillow `resize((224, 224), BICUBIC)` (antialiased:
    Pillow's two-pass convolution with 22-bit fixed-point coefficients and an 8-bit intermediate
    image), centre crop, /255, normalise (pipeline :153);
  * the VAE input: `to_tensor` (k/255 in float32) -> model dtype -> `* 2 - 1` (pipeline :169-170).
`pil_resize_bicubic_u8` restates Pillow's resampler (src/libImaging/Resample.c: precompute_coeffs,
normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc, bicubic a = -0.5) and is
pinned bit for bit to the installed Pillow in tests/test_mv_preprocess.py.
"""
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    """Resample.c lanczos_filter: truncated sinc, support 3."""
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


_FILTERS = {"bicubic": (_bicubic, 2.0), "lanczos": (_lanczos, 3.0)}


def _coeff_matrix(in_size, out_size, filt="bicubic"):
    """(out_size, in_size) int64 matrix of Pillow's normalised 22-bit coefficients."""
    _bicubic, support = _FILTERS[filt]          # (shadows the module-level name on purpose)
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    sup = support * filterscale
    ss = 1.0 / filterscale
    m = np.zeros((out_size, in_size), np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - sup + 0.5), 0)
        xmax = min(int(center + sup + 0.5), in_size)
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax - xmin)]
        ww = 0.0
        for v in w:
            ww += v
        for i, v in enumerate(w):
            k = v / ww if ww != 0.0 else v
            m[xx, xmin + i] = int(k * (1 << PRECISION_BITS) + (-0.5 if k < 0 else 0.5))
    return m


_COEFF_CACHE = {}


def _coeffs(in_size, out_size, device, filt="bicubic"):
    key = (in_size, out_size, str(device), filt)
    if key not in _COEFF_CACHE:
        t = torch.from_numpy(_coeff_matrix(in_size, out_size, filt)).to(device)
        from .._lib import publish_sync
        publish_sync(device)
        _COEFF_CACHE[key] = t
    return _COEFF_CACHE[key]


def _pass(img, m):
    """one resampling pass along dim 1 of (A, n_in, C) uint8 -> (A, n_out, C) uint8."""
    acc = torch.einsum("oi,aic->aoc", m.double(), img.double())      # |sum| < 2^31: exact in f64
    acc = acc.to(torch.int64) + (1 << (PRECISION_BITS - 1))
    return (acc >> PRECISION_BITS).clamp_(0, 255).to(torch.uint8)


def pil_resize_u8(img, out_hw, filt="bicubic"):
    """img (H, W, C) uint8 tensor (host or device) -> (out_h, out_w, C) uint8, = PIL.Image.resize((w,
    h), BICUBIC | LANCZOS) of an "RGB" / "L" image, bit for bit (tests/test_mv_preprocess.py)."""
    assert img.dtype == torch.uint8 and img.dim() == 3
    H, W, _ = img.shape
    oh, ow = out_hw
    x = img
    if ow != W:                                                       # horizontal pass first
        x = _pass(x.permute(0, 1, 2), _coeffs(W, ow, img.device, filt))
    if oh != H:
        x = _pass(x.permute(1, 0, 2), _coeffs(H, oh, img.device, filt)).permute(1, 0, 2)
    return x.contiguous()


def pil_resize_bicubic_u8(img, out_hw):
    return pil_resize_u8(img, out_hw, "bicubic")


def _muldiv255(a, b):
    """ImagingUtils.h MULDIV255 on int32 tensors: (a b + 128 + ((a b + 128) >> 8)) >> 8."""
    t = a * b + 128
    return (t + (t >> 8)) >> 8


def pil_resize_rgba_u8(img, out_hw, filt="bicubic"):
    """PIL.Image.resize of an "RGBA" image (Image.py: for RGBA / LA and any filter but NEAREST the
    image is converted to premultiplied "RGBa", resampled, and converted back):
    Convert.c rgbA2rgba = MULDIV255(c, a) per colour channel; the four channels through the
    resampler; rgba2rgbA = CLIP8(255 c / a) (integer division), colour untouched where a is 0.
    img (H, W, 4) uint8 -> (out_h, out_w, 4) uint8."""
    assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 4
    x = img.to(torch.int32)
    a = x[..., 3:4]
    pre = torch.cat([_muldiv255(x[..., :3], a), a], -1).to(torch.uint8)
    r = pil_resize_u8(pre, out_hw, filt).to(torch.int32)
    ra = r[..., 3:4]
    col = torch.where(ra > 0, torch.div(255 * r[..., :3], ra.clamp(min=1), rounding_mode="floor")
                      .clamp(max=255), r[..., :3])
    return torch.cat([col, ra], -1).to(torch.uint8)


def to_pil_u8(images):
    """torchvision to_pil_image on a float tensor (B,3,H,W): mul(255) in the tensor's dtype, then
    .byte() (truncation).  Returns (B,H,W,3) uint8."""
    assert images.is_floating_point()
    return images.mul(255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def clip_pixel_values(img_u8, size=224):
    """CLIPImageProcessor on (B,H,W,3) uint8 square images: resize to `size` (shortest edge; the
    centre crop is then the identity), rescale by 1/255, normalise.  Returns (B,3,size,size) f32."""
    assert img_u8.shape[1] == img_u8.shape[2], "square inputs (SingleImageDataset pads to a square)"
    out = torch.stack([pil_resize_bicubic_u8(im, (size, size)) for im in img_u8])
    x = out.permute(0, 3, 1, 2).float() * (1.0 / 255.0)
    mean = torch.tensor(CLIP_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std


def vae_input(img_u8, dtype):
    """to_tensor (k/255 in f32) -> dtype -> *2 - 1 in that dtype.  (B,H,W,3) u8 -> (B,3,H,W)."""
    x = (img_u8.permute(0, 3, 1, 2).float() / 255.0).to(dtype)
    return x * 2.0 - 1.0


__all__ = ["pil_resize_u8", "pil_resize_rgba_u8", "pil_resize_bicubic_u8", "to_pil_u8", "clip_pixel_values", "vae_input", "math"]
