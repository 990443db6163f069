"""torch.autograd.Function wrappers that put the gfx950 style-translator kernels under autograd
for per-character training (3_style_translator/training/trainers.py:140-192 runs cuDNN /
torchvision backward passes here).  Every forward AND backward is a libdsu_hip kernel; torch
only links the nodes (and owns cat / add / views).  Nothing here runs on a CPU tensor.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import ops


class ConvFn(Function):
    """nn.Conv2d (plan None) or torchvision.ops.deform_conv2d with the fixed offsets of `plan`
    (3x3, stride 1, padding 1), optionally followed by ReLU / LeakyReLU(0.2) / tanh fused into
    the convolution epilogue."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, act, plan):
        x = x.contiguous()
        if plan is not None:
            assert bias is None and stride == 1 and padding == 1
            y = ops.deform_conv3x3(x, plan.offset, weight, act=act)
        else:
            y = ops.conv2d(x, weight, bias, stride, padding, act=act)
        ctx.cfg = (stride, padding, act, plan, bias is not None)
        ctx.save_for_backward(x, weight, y if act else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        stride, padding, act, plan, has_bias = ctx.cfg
        dy = dy.contiguous()
        if act:
            dy = ops.act_bwd(dy, y, act)
        k = weight.shape[2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if plan is not None:
                dx = ops.deform_conv3x3_dgrad(dy, weight, plan)
            else:
                dx = ops.conv2d_dgrad(dy, weight, x.shape[2:], stride, padding)
        if ctx.needs_input_grad[1]:
            dw = ops.conv2d_wgrad(x, dy, k, stride, padding, plan)
        if has_bias and ctx.needs_input_grad[2]:
            db = ops.channel_sum(dy)
        return dx, dw, db, None, None, None, None


class NormFn(Function):
    """nn.BatchNorm2d in training mode (running statistics updated in place by the kernel) or
    nn.InstanceNorm2d(affine=False), with the following ReLU / LeakyReLU(0.2) fused."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, instance, act, eps, momentum,
                stat_updates):
        x = x.contiguous()
        y, mean, invstd = ops.norm_train_fwd(x, gamma, beta, running_mean, running_var, instance,
                                             act, eps, momentum, stat_updates)
        ctx.cfg = (instance, act)
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, y, gamma, mean, invstd = ctx.saved_tensors
        instance, act = ctx.cfg
        affine = gamma is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        dx, dg, db = ops.norm_train_bwd(x, y, dy.contiguous(), gamma, mean, invstd, instance, act,
                                        affine)
        return dx, dg, db, None, None, None, None, None, None, None


class ActFn(Function):
    @staticmethod
    def forward(ctx, x, act):
        y = ops.act_fwd(x.contiguous(), act)
        ctx.act = act
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.act_bwd(dy.contiguous(), y, ctx.act), None


class MaxPool2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.maxpool2_fwd(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool2_bwd(x, dy.contiguous())


class Upsample2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        return ops.upsample2_fwd(x.contiguous())

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return ops.upsample2_bwd(dy.contiguous())


class PairLossFn(Function):
    """mean |x - t| (kind 'l1') or mean (x - t)^2 (kind 'mse'); t a tensor (no gradient) or a
    float.  The gradient is produced by the same kernel launch as the value."""

    @staticmethod
    def forward(ctx, x, target, kind):
        x = x.contiguous()
        n = x.numel()
        if torch.is_tensor(target):
            target = target.detach().contiguous()
            assert target.shape == x.shape, (target.shape, x.shape)
        total, grad = ops.pair_loss(x, target, kind, grad_scale=1.0 / n)
        ctx.save_for_backward(grad)
        return total / n

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        (grad,) = ctx.saved_tensors
        return grad * dloss, None, None


def conv(x, weight, bias=None, stride=1, padding=0, act=None, plan=None):
    return ConvFn.apply(x, weight, bias, stride, padding, act, plan)


def batch_norm_train(x, bn, act=None, stat_updates=1):
    """bn: nn.BatchNorm2d (affine, track_running_stats).  momentum None (cumulative average) is
    not something the reference configures."""
    assert bn.momentum is not None and bn.affine and bn.track_running_stats
    y = NormFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, False, act, bn.eps,
                     bn.momentum, stat_updates)
    bn.num_batches_tracked += stat_updates
    return y


def instance_norm(x, act=None, eps=1e-5):
    return NormFn.apply(x, None, None, None, None, True, act, eps, 0.0, 0)


def activation(x, act):
    return ActFn.apply(x, act)


def maxpool2(x):
    return MaxPool2Fn.apply(x)


def upsample2(x):
    return Upsample2Fn.apply(x)


def l1_loss(x, target):
    return PairLossFn.apply(x, target, "l1")


def mse_loss(x, target):
    return PairLossFn.apply(x, target, "mse")
