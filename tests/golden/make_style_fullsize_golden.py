"""Generate tests/golden/style_fullsize_reference.npz: the REFERENCE's own GeneratorJ and
GeneratorJ_RIC (3_style_translator/training/models.py) at the SHIPPED widths
(configs/config_stage{1,2}.yaml: filters [32,64,128,128,128,64], 7 res-blocks) on one 512x512
frame, on the CPU in this container.

    python tests/golden/make_style_fullsize_golden.py     # needs /root/reference (~10 min)

The fixture stores SEEDS, not weights: parameters come from oracle.style_ref.seeded_state_dict
(the test rebuilds the same state_dict), the input from a seeded generator.  Stored per
generator: the uint8 image (custom_transforms.py:8-9 `to_image_space`) at full resolution and
the float32 output on a stride-3 lattice.  GeneratorJ is the reference end to end;
GeneratorJ_RIC runs the reference graph with torchvision.ops.deform_conv2d supplied by the
oracle restatement (op unpinned, graph pinned) — as in make_style_golden.py.
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

ARGS, frame = style_ref.FULLSIZE_ARGS, style_ref.fullsize_frame
SEEDS = {"GeneratorJ": 101, "GeneratorJ_RIC": 202}
STRIDE = 3


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 1)
    out = {"stride": np.int64(STRIDE)}
    for name, seed in SEEDS.items():
        net = getattr(ref_models, name)(**ARGS)
        net.load_state_dict(style_ref.seeded_state_dict(net.state_dict(), seed))
        net.eval()
        x = frame(seed + 1)
        with torch.no_grad():
            y = net(x)[0].numpy()
        out[name + ".seed"] = np.int64(seed)
        out[name + ".u8"] = ((np.clip(y, -1, 1) + 1) / 2 * 255).astype(np.uint8)
        out[name + ".f32"] = y[:, ::STRIDE, ::STRIDE].copy()
        print(name, "done: mean |y|", flush=True) if False else print(name, "done: mean |y|", float(np.abs(y).mean()), "saturated", float((np.abs(y) > 0.999).mean()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "style_fullsize_reference.npz"), **out)
    print("wrote style_fullsize_reference.npz")
