"""Generate tests/golden/style_reference.npz by running the REFERENCE's own
3_style_translator/training/models.py (GeneratorJ, GeneratorJ_RIC, generate_coordinates) on
the CPU in this container.

    python tests/golden/make_style_golden.py        # needs /root/reference (not on the GPU box)

torchvision is not installable here, so `torchvision.ops.deform_conv2d` is provided by the
oracle's restatement (oracle/style_ref.py) — the fixture therefore pins the reference's GRAPH
(layer order, BN placement, the dead smoother conv, skip connections, offset map) and is
"parity unpinned" only for the deform_conv2d op itself.  GeneratorJ uses no third-party op:
its fixture is the reference end to end.  Reduced widths keep the fixture small; the layer
structure is the shipped config's (configs/config_stage{1,2}.yaml).
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/3_style_translator"

from oracle import style_ref  # noqa: E402

tv = types.ModuleType("torchvision")
tv.ops = types.ModuleType("torchvision.ops")
tv.models = types.ModuleType("torchvision.models")
tv.ops.deform_conv2d = lambda input, offset, weight, padding=(1, 1): \
    style_ref.deform_conv2d(input, offset, weight, padding).to(input.dtype)
sys.modules["torchvision"] = tv
sys.modules["torchvision.ops"] = tv.ops
sys.modules["torchvision.models"] = tv.models
torch.Tensor.cuda = lambda self, *a, **k: self      # generate_coordinates hard-codes .cuda()
sys.path.insert(0, REF)
from training import models as ref_models  # noqa: E402


def run(cls_name, seed, H, W):
    torch.manual_seed(seed)
    args = dict(use_bias=False, tanh=True, append_smoothers=True, resnet_blocks=2,
                filters=[8, 16, 24, 24, 24, 16], input_channels=6)
    net = getattr(ref_models, cls_name)(**args)
    # non-trivial BatchNorm statistics (eval mode uses them)
    g = torch.Generator().manual_seed(seed + 1)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    net.eval()
    x = torch.rand(1, 6, H, W, generator=g) * 2 - 1
    with torch.no_grad():
        y = net(x)
    sd = {f"{cls_name}.sd.{k}": v.numpy() for k, v in net.state_dict().items()}
    return {f"{cls_name}.x": x.numpy(), f"{cls_name}.y": y.numpy(), **sd}


out = {}
out.update(run("GeneratorJ", 10, 32, 40))
out.update(run("GeneratorJ_RIC", 20, 32, 40))
for (H, W) in [(16, 16), (32, 40), (64, 48)]:
    out[f"coords.{H}x{W}"] = ref_models.generate_coordinates(1, H, W)[0].numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "style_reference.npz"), **out)
print("wrote style_reference.npz:", {k: v.shape for k, v in out.items() if ".sd." not in k})
