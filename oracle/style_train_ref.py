"""CPU oracle for the style-translator TRAINING ops.  TEST INFRASTRUCTURE ONLY.

Torch (CPU) restatements of what the reference's training graph executes through autograd
(3_style_translator/training/trainers.py:87-137 over models.py:113-129, 293-356, 426-477,
536-549): each function has the signature of its drawingspinup_amd.style.functions counterpart,
so tests can (a) check a HIP kernel against it and (b) run the product's modules and Trainer on
the CPU with these stand-ins to pin the graph WIRING against the reference's own trainer fixture
(tests/golden/style_train_reference.npz).  Pinned by that fixture; the deformable convolution
itself is oracle/style_ref.py (parity unpinned for that op alone, see its header).
"""
import types

import torch
import torch.nn.functional as F

from . import style_ref

ACT = {None: lambda t: t, "relu": F.relu, "leaky_relu": lambda t: F.leaky_relu(t, 0.2),
       "tanh": torch.tanh}


def conv(x, weight, bias=None, stride=1, padding=0, act=None, plan=None):
    """nn.Conv2d (models.py:41-129, 441-462; VGG19 features) or, with `plan`,
    torchvision.ops.deform_conv2d(padding=(1,1)) with the fixed offsets plan.offset
    (models.py:302-351), followed by the activation the product fuses."""
    if plan is not None:
        off = plan.offset[None].expand(x.shape[0], -1, -1, -1)
        y = style_ref.deform_conv2d(x, off, weight).to(x.dtype)
    else:
        y = F.conv2d(x, weight, bias, stride, padding)
    return ACT[act](y)


def batch_norm_train(x, bn, act=None, stat_updates=1):
    """nn.BatchNorm2d in training mode; `stat_updates` forwards' worth of running-stat updates
    (the reference runs the generator twice per iteration, trainers.py:88,102)."""
    y = F.batch_norm(x, None, None, bn.weight, bn.bias, True, bn.momentum, bn.eps)
    with torch.no_grad():
        for _ in range(stat_updates):
            F.batch_norm(x, bn.running_mean, bn.running_var, None, None, True, bn.momentum, bn.eps)
        bn.num_batches_tracked += stat_updates
    return ACT[act](y)


def instance_norm(x, act=None, eps=1e-5):
    """nn.InstanceNorm2d(affine=False) of DiscriminatorN_IN (models.py:436-439) + LeakyReLU."""
    return ACT[act](F.instance_norm(x, eps=eps))


def activation(x, act):
    return ACT[act](x)


def maxpool2(x):
    """nn.MaxPool2d(2, 2) (models.py:214; VGG19 features[4])."""
    return F.max_pool2d(x, 2, 2)


def upsample2(x):
    """nn.Upsample(scale_factor=2) (models.py:8-14)."""
    return F.interpolate(x, scale_factor=2)


def _loss(f):
    def loss(x, target):
        if not torch.is_tensor(target):
            target = torch.full_like(x, float(target))
        return f(x, target.detach())
    return loss


l1_loss = _loss(F.l1_loss)      # nn.L1Loss: reconstruction_criterion (config_stage1.yaml:56)
mse_loss = _loss(F.mse_loss)    # nn.MSELoss: adversarial_criterion; ((a-b)**2).mean() trainers.py:128


def deform_plan(offset):
    """Stand-in for ops.deform_plan: the oracle samples from the offsets directly."""
    return types.SimpleNamespace(offset=offset)
