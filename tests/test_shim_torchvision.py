"""The `torchvision` drop-in must not take `torchvision.models` / `torchvision.transforms` away
from the reference's own files: 3_style_translator/training/models.py:4-5 and
custom_transforms.py:2 are imported UNMODIFIED with the shims first on sys.path (CPU, build
container only — the snapshot is not on the GPU box)."""
import os
import subprocess
import sys

import pytest

REF = "/root/reference/3_style_translator"
CODE = r'''
import os, sys, types
sys.path.insert(0, %(root)r)
from drawingspinup_amd import shims
shims.install()
sys.modules["cv2"] = types.ModuleType("cv2")            # custom_transforms imports it; unused here
sys.path.insert(0, %(ref)r)
import torchvision
from training import models as M                        # `import torchvision; from torchvision import models`
from training import custom_transforms as CT            # `from torchvision import transforms`
from training import data as D
assert hasattr(torchvision, "ops") and hasattr(torchvision.ops, "deform_conv2d")
assert hasattr(torchvision, "models") and hasattr(torchvision.models, "vgg19")
assert all(hasattr(torchvision.transforms, n) for n in ("Compose", "ToTensor", "Normalize"))
from drawingspinup_amd.style.generators import deform_conv2d
assert torchvision.ops.deform_conv2d is deform_conv2d
import numpy as np
from PIL import Image
t = CT.build_transform()(Image.fromarray(np.full((4, 5, 4), 255, np.uint8), "RGBA"))
assert tuple(t.shape) == (3, 4, 5) and float(t.min()) == 1.0
m = CT.build_mask_transform()(Image.fromarray(np.full((4, 5), 128, np.uint8), "L"))
assert tuple(m.shape) == (1, 4, 5) and abs(float(m[0, 0, 0]) - 128 / 255) < 1e-7
g = M.GeneratorJ_RIC(use_bias=False, tanh=True, append_smoothers=True, resnet_blocks=1,
                     filters=[8, 8, 8, 8, 8, 8], input_channels=6)     # the reference's class builds
vgg = torchvision.models.vgg19(pretrained=False)
assert len(vgg.features) == 37 and vgg.features[34].weight.shape == (512, 512, 3, 3)
try:
    torchvision.models.vgg19(pretrained=True)
    raise SystemExit("pretrained=True without a weights file must fail")
except FileNotFoundError:
    pass
print("torchvision shim ok")
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference snapshot is only in the build container")
def test_reference_style_files_import_under_the_shim(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TORCH_HOME=str(tmp_path))
    env.pop("DSU_VGG19_WEIGHTS", None)
    r = subprocess.run([sys.executable, "-c", CODE % {"root": root, "ref": REF}], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0 and "torchvision shim ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
