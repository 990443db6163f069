"""CPU: the style oracle / host glue against the fixture produced by the reference's own code."""
import os

import numpy as np
import torch

from drawingspinup_amd.style import generators as G
from oracle import style_ref as sr

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "style_reference.npz"))


def test_generate_coordinates_bit_exact_vs_reference():
    for key in [k for k in GOLD.files if k.startswith("coords.")]:
        H, W = map(int, key.split(".")[1].split("x"))
        ref = torch.from_numpy(GOLD[key])
        assert torch.equal(sr.generate_coordinates(H, W), ref)                  # oracle
        host = G.generate_coordinates(2, H, W, device="cpu")                    # product host glue
        assert host.shape == (2, 18, H, W) and torch.equal(host[1], ref)
        assert torch.equal(ref[8:10], torch.zeros(2, H, W))                      # centre tap KAT


def test_state_dict_keys_match_reference():
    args = dict(use_bias=False, tanh=True, append_smoothers=True, resnet_blocks=2,
                filters=[8, 16, 24, 24, 24, 16], input_channels=6)
    for name in ("GeneratorJ", "GeneratorJ_RIC"):
        ref_keys = sorted(k.split(".sd.")[1] for k in GOLD.files if k.startswith(name + ".sd."))
        net = G.build_model(name, args)
        assert sorted(net.state_dict().keys()) == ref_keys
        sd = {k: torch.from_numpy(GOLD[f"{name}.sd.{k}"]) for k in ref_keys}
        net.load_state_dict(sd)          # strict: shapes and names
    # the keys SURVEY.md §5 lists for the shipped checkpoints
    assert "conv0.conv.weight" in ref_keys and "conv_11_a.3.weight" in ref_keys \
        and "conv_12.0.bias" in ref_keys and "resnets.1.conv_0.weight" in ref_keys


def test_oracle_deform_conv_kats():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 9, 11, generator=g)
    w = torch.randn(7, 5, 3, 3, generator=g)
    # zero offsets == plain convolution
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    assert torch.allclose(sr.deform_conv2d(x, torch.zeros(2, 18, 9, 11), w), ref, atol=1e-12)
    # integer offset (+1 row on every tap) == convolution of the image shifted up by one row
    off = torch.zeros(2, 18, 9, 11)
    off[:, 0::2] = 1.0
    xs = torch.zeros_like(x)
    xs[:, :, :-1] = x[:, :, 1:]
    ref2 = torch.nn.functional.conv2d(xs.double(), w.double(), padding=1)
    got2 = sr.deform_conv2d(x, off, w)
    # identical wherever the shifted image's zero padding is not involved (output rows >= 1;
    # row 0's top taps read real row 0 in the deformable op but padding in the shifted conv)
    assert torch.allclose(got2[:, :, 1:], ref2[:, :, 1:], atol=1e-12)
    # half-pixel offset averages two neighbours
    off3 = torch.zeros(1, 18, 4, 4)
    off3[:, 1::2] = 0.5
    x3 = torch.arange(16.0).view(1, 1, 4, 4)
    w3 = torch.zeros(1, 1, 3, 3); w3[0, 0, 1, 1] = 1.0
    got3 = sr.deform_conv2d(x3, off3, w3)[0, 0]
    exp3 = (x3[0, 0] + torch.cat([x3[0, 0][:, 1:], torch.zeros(4, 1)], 1)) * 0.5
    assert torch.allclose(got3, exp3.double())


def test_generators_refuse_cpu_and_train_mode():
    args = dict(use_bias=False, tanh=True, append_smoothers=True, resnet_blocks=1,
                filters=[8, 8, 8, 8, 8, 8], input_channels=6)
    net = G.build_model("GeneratorJ", args)
    x = torch.zeros(1, 6, 8, 8)
    import pytest
    with pytest.raises(RuntimeError):
        net(x)                      # train mode
    net.eval()
    with pytest.raises(RuntimeError):
        net(x)                      # CPU tensor: no fallback
