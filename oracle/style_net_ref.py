"""GeneratorJ.forward of the reference (3_style_translator/training/models.py:113-129) executed
with the module tree's OWN torch sub-modules on the CPU.  TEST INFRASTRUCTURE ONLY (CPU baseline
of bench.py and its pinning test).

drawingspinup_amd.style.generators.GeneratorJ builds the reference's module tree (same
sub-module names and state_dict keys) but its forward dispatches to the HIP kernels; this function
walks the same tree through nn.Conv2d / nn.BatchNorm2d / activations — the exact operator sequence
the reference's class runs, i.e. what the reference costs on host cores.  Pinned by
tests/test_oracle_style_net.py against the reference-generated fixture
(tests/golden/style_reference.npz)."""
import torch


def generator_j_forward(net, x):
    o0 = net.conv0(x)
    o1 = net.conv1(o0)
    o2 = net.conv2(o1)
    out = o2
    for layer in net.resnets:
        out = layer(out) + out
    out = net.upconv2(torch.cat((out, o2), dim=1))
    out = net.upconv1(torch.cat((out, o1), dim=1))
    out = net.conv_11(torch.cat((out, o0, x), dim=1))
    if net.append_smoothers:
        out = net.conv_11_a(out)
    return net.conv_12(out)
