"""`torchvision.models.vgg19` as 3_style_translator/training/models.py:485,497 uses it
(PerceptualVGG19 reads `.features` and drops `.classifier`).  The published configuration "E":
16 3x3 convolutions + ReLU, 5 max-pools; parameter names `features.N.{weight,bias}` as in the
torchvision checkpoint.

`pretrained=True` needs the ImageNet file; there is no network here, so it is read from
DSU_VGG19_WEIGHTS or the torch hub cache (~/.cache/torch/hub/checkpoints/vgg19-*.pth) and a
missing file is an ERROR (the perceptual loss on random features trains a different model)."""
import glob
import os

import torch
import torch.nn as nn

_CFG_E = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
          512, 512, 512, 512, "M"]


class VGG(nn.Module):
    def __init__(self):
        super().__init__()
        layers, c = [], 3
        for v in _CFG_E:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                c = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(
            nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(),
            nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 1000))

    def forward(self, x):
        x = self.avgpool(self.features(x))
        return self.classifier(torch.flatten(x, 1))


def find_vgg19_weights():
    cand = [os.environ.get("DSU_VGG19_WEIGHTS")]
    hub = os.path.join(os.environ.get("TORCH_HOME", os.path.expanduser("~/.cache/torch")), "hub",
                       "checkpoints")
    cand += sorted(glob.glob(os.path.join(hub, "vgg19-*.pth")))
    for c in cand:
        if c and os.path.isfile(c):
            return c
    return None


def vgg19(pretrained=False, weights=None, **kw):
    m = VGG()
    if pretrained or weights is not None:
        path = find_vgg19_weights()
        if path is None:
            raise FileNotFoundError(
                "vgg19(pretrained=True): no ImageNet weights file (no network in this environment); "
                "set DSU_VGG19_WEIGHTS=/path/to/vgg19-dcbb9e9d.pth or place it in the torch hub cache")
        m.load_state_dict(torch.load(path, map_location="cpu"))
    return m
