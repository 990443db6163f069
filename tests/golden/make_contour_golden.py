"""Generate tests/golden/contour_reference.npz by running the REFERENCE's own FFC-ResNet
(1_lama_contour_remover/saicinpainting/training/modules/ffc.py, `make_generator`) on the CPU in
this container.

    python tests/golden/make_contour_golden.py        # needs /root/reference (not on the GPU box)

The reference module is imported as is (kornia, only used by an option the shipped config does
not enable, is stubbed), so this fixture pins the contour-remover generator to the reference
end to end: the oracle IS the reference.  Two things are stored:
  * a reduced-width instance (ngf 8, 2 blocks, otherwise configs/prediction/lama-fourier.yaml):
    state_dict, one input, its output -> value parity and state_dict-key parity;
  * the (name, shape) list of the full-size shipped configuration (27.04 M parameters) -> key /
    shape parity of the real checkpoint layout.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/1_lama_contour_remover"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "contour_reference.npz")

for name in ("kornia", "kornia.geometry", "kornia.geometry.transform"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["kornia.geometry.transform"].rotate = None
sys.path.insert(0, REF)
from saicinpainting.training.modules import make_generator  # noqa: E402

FULL = dict(kind="ffc_resnet", input_nc=4, output_nc=1, ngf=64, n_downsampling=3, n_blocks=9,
            add_out_act="sigmoid",
            init_conv_kwargs=dict(ratio_gin=0, ratio_gout=0, enable_lfu=False),
            downsample_conv_kwargs=dict(ratio_gin=0, ratio_gout=0, enable_lfu=False),
            resnet_conv_kwargs=dict(ratio_gin=0.75, ratio_gout=0.75, enable_lfu=False))
SMALL = dict(FULL, ngf=8, n_blocks=2)


def main():
    torch.manual_seed(0)
    full = make_generator(**FULL)
    keys = [(k, tuple(v.shape)) for k, v in full.state_dict().items()]
    torch.manual_seed(1)
    small = make_generator(**SMALL).eval()
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for m in small.modules():                      # non-trivial BN statistics
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
        x = torch.rand(2, 4, 48, 40, generator=g)       # 48x40: non-square, 6x5 at the bottleneck
        y = small(x)
    data = {"x": x.numpy(), "y": y.numpy(),
            "full_keys": np.array([k for k, _ in keys]),
            "full_shapes": np.array([",".join(map(str, s)) for _, s in keys]),
            "full_params": np.array(sum(p.numel() for p in full.parameters()))}
    for k, v in small.state_dict().items():
        data["sd/" + k] = v.numpy()
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB;", len(keys), "full-size keys;",
          int(data["full_params"]), "parameters; y range", float(y.min()), float(y.max()))


if __name__ == "__main__":
    main()
