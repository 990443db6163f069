import numpy as np
import torch
from PIL import Image


def to_pil_image(pic, mode=None):
    """tensor (C,H,W): float -> `pic.mul(255).byte()` (in the tensor's dtype, truncating)."""
    assert isinstance(pic, torch.Tensor) and pic.ndimension() == 3 and pic.shape[0] == 3 and mode is None
    if pic.is_floating_point():
        pic = pic.mul(255).byte()
    npimg = np.transpose(pic.cpu().numpy(), (1, 2, 0))
    assert npimg.dtype == np.uint8
    return Image.fromarray(npimg, mode="RGB")


def to_tensor(pic):
    """PIL RGB -> float32 (C,H,W) in [0,1] (uint8 / 255)."""
    img = torch.from_numpy(np.array(pic, np.uint8, copy=True))
    img = img.view(pic.size[1], pic.size[0], len(pic.getbands())).permute(2, 0, 1).contiguous()
    return img.to(dtype=torch.float32).div(255)
