"""oracle/style_net_ref.generator_j_forward (bench.py's CPU-baseline leg for the style nets)
against the fixture produced by the reference's own GeneratorJ class."""
import os

import numpy as np
import torch

from drawingspinup_amd.style import generators as G
from oracle import style_net_ref as snr

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "style_reference.npz"))


def test_cpu_generator_j_forward_matches_reference_fixture():
    args = dict(use_bias=False, tanh=True, append_smoothers=True, resnet_blocks=2,
                filters=[8, 16, 24, 24, 24, 16], input_channels=6)
    net = G.build_model("GeneratorJ", args, "cpu")
    net.load_state_dict({k.split(".sd.")[1]: torch.from_numpy(GOLD[k]) for k in GOLD.files
                         if k.startswith("GeneratorJ.sd.")})
    net.eval()
    with torch.no_grad():
        y = snr.generator_j_forward(net, torch.from_numpy(GOLD["GeneratorJ.x"]))
    np.testing.assert_allclose(y.numpy(), GOLD["GeneratorJ.y"], rtol=0, atol=1e-6)
