"""`torchvision.transforms.{Compose, ToTensor, Normalize}` as
3_style_translator/training/custom_transforms.py:17-27 uses them (published semantics)."""
import numpy as np
import torch


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class ToTensor:
    """PIL image / uint8 HWC array -> float32 CHW in [0, 1]."""

    def __call__(self, pic):
        a = np.array(pic)
        if a.ndim == 2:
            a = a[..., None]
        t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).contiguous()
        return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t.to(torch.float32)


class Normalize:
    def __init__(self, mean, std, inplace=False):
        self.mean, self.std = mean, std

    def __call__(self, t):
        mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        return (t - mean) / std
