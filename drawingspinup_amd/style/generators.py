"""GeneratorJ / GeneratorJ_RIC with the reference's constructor, module tree and state_dict
keys (3_style_translator/training/models.py:24-192, 200-356), evaluated with the gfx950
f32-MFMA convolution kernels.

eval() (the reference's test_stage1.py / test_stage2.py path): BatchNorm is folded into the
convolution epilogue together with the activation; the resnet blocks' leading ReLU is applied
as the input is read; the residual add is fused.  No autograd graph is built.

train() (train_stage1.py / train_stage2.py): the same module tree evaluated
op by op through style/functions.py — batch-statistics BatchNorm with running-stat updates,
every forward and backward a libdsu_hip kernel under torch.autograd.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from . import functions as Fn


def deform_conv2d(input, offset, weight, bias=None, stride=(1, 1), padding=(0, 0),
                  dilation=(1, 1), mask=None):
    """torchvision.ops.deform_conv2d for the configuration the reference uses
    (models.py:302-351): 3x3 weight, stride 1, padding (1,1), no bias / mask."""
    def _pair(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    if _pair(stride) != (1, 1) or _pair(padding) != (1, 1) or _pair(dilation) != (1, 1) \
            or mask is not None or tuple(weight.shape[2:]) != (3, 3):
        raise NotImplementedError("gfx950 deform_conv2d: 3x3, stride 1, padding 1, no mask")
    off = offset
    if off.dim() == 4 and off.shape[0] > 1 and off.stride(0) == 0:
        off = off[0]          # the reference expands one map over the batch (models.py:600)
    out = ops.deform_conv3x3(input, off.contiguous(), weight)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


_COORD_CACHE = {}


def generate_coordinates(batch_size, input_height, input_width, device="cuda"):
    """models.py:551-604, same torch op sequence on the host (so the 1e-4 angle rounding is
    bit-identical to the reference), cached per (H, W), uploaded once.  Returns
    (batch, 18, H, W) as an expanded view like the reference."""
    key = (input_height, input_width, str(device))
    if key not in _COORD_CACHE:
        H, W = input_height, input_width
        coords = torch.zeros(H, W, 18)
        p = torch.zeros(3)
        p[0], p[1], p[2] = batch_size, H, W
        center = torch.zeros(2)
        center[0] = torch.sub(torch.div(p[1], 2.0), 0.5)
        center[1] = torch.sub(torch.div(p[2], 2.0), 0.5)
        grid_x, grid_y = torch.meshgrid(torch.arange(0, p[1]), torch.arange(0, p[2]),
                                        indexing="ij")
        delta_x, delta_y = torch.sub(grid_x, center[0]), torch.sub(grid_y, center[1])
        PI = torch.mul(torch.Tensor([math.pi]), 2.0)
        theta = torch.atan2(delta_y, delta_x) % PI[0]
        theta = torch.round(10000. * theta) / 10000.
        shift = [(1., 1.), (1., 0.), (1., -1.), (0., 1.), None, (0., -1.), (-1., 1.), (-1., 0.),
                 (-1., -1.)]
        for k in range(9):
            if k == 4:
                continue
            m = float(k if k < 4 else k - 1)
            ang = torch.add(theta, torch.mul(torch.div(PI[0], 8.0), m))
            coords[:, :, 2 * k] = torch.add(torch.cos(ang), shift[k][0])
            coords[:, :, 2 * k + 1] = torch.add(torch.sin(ang), shift[k][1])
        t = coords.permute(2, 0, 1).contiguous().to(device)
        from .._lib import publish_sync
        publish_sync(device)
        _COORD_CACHE[key] = t
    c = _COORD_CACHE[key]
    return c.unsqueeze(0).expand(batch_size, -1, -1, -1)


class UpsamplingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.layer = nn.Upsample(scale_factor=2)

    def forward(self, x):
        return self.layer(x)


def _bn_fold(bn):
    """eval BatchNorm2d -> per-channel (scale, shift), cached on the module."""
    ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.device)
    cache = getattr(bn, "_dsu_fold", None)
    if cache is None or cache[0] != ver:
        scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
        shift = (bn.bias - bn.running_mean * scale).float().contiguous()
        bn._dsu_fold = (ver, scale, shift)
        cache = bn._dsu_fold
    return cache[1], cache[2]


# Evaluation arithmetic of the generators' convolutions, chosen per operator as the reference's own
# arithmetic allows (module attributes for the A/B tools and tests, not deployment switches):
#   EVAL_X3         plain nn.Conv2d layers.  The reference runs them on cuDNN with
#                   torch.backends.cudnn.allow_tf32 = True (PyTorch's default): 10-bit mantissa
#                   products.  True = bf16 x 3 products on the bf16 MFMA (2^-15 relative per
#                   product, finer than TF32); False = exact f32 products.
#   EVAL_DEFORM_X3  torchvision.ops.deform_conv2d layers (every 3x3 layer of GeneratorJ_RIC,
#                   models.py:302-351).  torchvision forms the im2col matrix in f32 and multiplies
#                   with addmm; torch.backends.cuda.matmul.allow_tf32 is False by default, so the
#                   reference's products are exact f32.  False (default) = exact f32 products on the
#                   f32 MFMA; True = bf16 x 3 (within the stated tolerance, NOT the reference's
#                   arithmetic: an A/B switch only).
# Both arithmetics run in style_conv_x3.hip (im2col values formed in registers, packed weights);
# layers it does not cover (7x7 in exact f32, 4x4) use style_conv.hip, which training always uses.
EVAL_X3 = True
EVAL_DEFORM_X3 = False


def _packed(conv, exact):
    """The convolution's weight in the evaluation kernels' packed layout (bf16 hi/mid parts, or f32
    for the exact kernels), cached on the module (rebuilt when the parameter's version or storage
    changes; dropped by train())."""
    w = conv.weight
    ver = (w._version, w.data_ptr(), w.device, bool(exact))
    cache = getattr(conv, "_dsu_pack", None)
    if cache is None or cache[0] != ver:
        conv._dsu_pack = (ver, ops.PackedConvWeight(w, exact=exact))
        cache = conv._dsu_pack
    return cache[1]


def _cat_in(tensors):
    """Channel concatenation feeding a convolution, zero channels appended up to a multiple of eight
    (what the packed-weight kernels read; the packed weights of those channels are zero)."""
    return ops.cat_channels8(tensors)


def _packed_ok(conv, exact):
    k, s = conv.kernel_size[0], conv.stride[0]
    return (k, s) in (((1, 1), (3, 1), (3, 2)) if exact else ((1, 1), (3, 1), (3, 2), (7, 1)))


def _act_name(m):
    if m is None:
        return None
    if isinstance(m, nn.LeakyReLU):
        assert abs(m.negative_slope - 0.2) < 1e-12
        return "leaky_relu"
    if isinstance(m, nn.ReLU):
        return "relu"
    if isinstance(m, nn.Tanh):
        return "tanh"
    raise NotImplementedError(type(m))


class _GeneratorBase(nn.Module):
    RIC = False

    def __init__(self, norm_layer="batch_norm", gpu_ids=None, use_bias=False, resnet_blocks=9,
                 tanh=False, filters=(64, 128, 128, 128, 128, 64), input_channels=3,
                 append_smoothers=False):
        super().__init__()
        assert norm_layer in [None, "batch_norm"], "gfx950 path folds eval BatchNorm only"
        self.norm_layer = nn.BatchNorm2d if norm_layer == "batch_norm" else None
        self.gpu_ids, self.use_bias = gpu_ids, use_bias
        self.resnet_blocks, self.append_smoothers = resnet_blocks, append_smoothers
        k0, s12 = (3, 1) if self.RIC else (7, 2)
        if self.RIC:
            self.maxpool = nn.MaxPool2d(kernel_size=2, stride=2, padding=0)
        self.conv0 = self.relu_layer(input_channels, filters[0], k0, 1, k0 // 2, use_bias,
                                     self.norm_layer, nn.LeakyReLU(.2))
        self.conv1 = self.relu_layer(filters[0], filters[1], 3, s12, 1, use_bias,
                                     self.norm_layer, nn.LeakyReLU(.2))
        self.conv2 = self.relu_layer(filters[1], filters[2], 3, s12, 1, use_bias,
                                     self.norm_layer, nn.LeakyReLU(.2))
        self.resnets = nn.ModuleList()
        for _ in range(resnet_blocks):
            self.resnets.append(self.resnet_block(filters[2], filters[2], 3, 1, 1, use_bias,
                                                  self.norm_layer, nn.ReLU()))
        self.upconv2 = self.upconv_layer_upsample_and_conv(filters[3] + filters[2], filters[4],
                                                           use_bias, self.norm_layer, nn.ReLU())
        self.upconv1 = self.upconv_layer_upsample_and_conv(filters[4] + filters[1], filters[4],
                                                           use_bias, self.norm_layer, nn.ReLU())
        self.conv_11 = nn.Sequential(
            nn.Conv2d(filters[0] + filters[4] + input_channels, filters[5], kernel_size=k0,
                      stride=1, padding=k0 // 2, bias=use_bias), nn.ReLU())
        if append_smoothers:
            self.conv_11_a = nn.Sequential(
                nn.Conv2d(filters[5], filters[5], kernel_size=3, bias=use_bias, padding=1),
                nn.ReLU(), nn.BatchNorm2d(num_features=filters[5]),
                nn.Conv2d(filters[5], filters[5], kernel_size=3, bias=use_bias, padding=1),
                nn.ReLU())
        if tanh:
            self.conv_12 = nn.Sequential(nn.Conv2d(filters[5], 3, kernel_size=1, stride=1,
                                                   padding=0, bias=True), nn.Tanh())
        else:
            self.conv_12 = nn.Conv2d(filters[5], 3, kernel_size=1, stride=1, padding=0, bias=True)

    def train(self, mode=True):
        """Entering or leaving training drops the folded eval-BatchNorm constants: the training
        kernels update running statistics through raw pointers and the fused Adam step does not
        bump `Parameter._version`, so `_bn_fold`'s version check alone would keep serving the
        constants of the last evaluation (e.g. Trainer.test_on_full_image every log_interval)."""
        for m in self.modules():
            for attr in ("_dsu_fold", "_dsu_pack"):
                if hasattr(m, attr):
                    delattr(m, attr)
        return super().train(mode)

    # ---- constructors with the reference's sub-module names (state_dict keys)
    @staticmethod
    def relu_layer(in_filters, out_filters, size, stride, padding, bias, norm_layer, nonlinearity):
        out = nn.Sequential()
        out.add_module("conv", nn.Conv2d(in_filters, out_filters, kernel_size=size, stride=stride,
                                         padding=padding, bias=bias))
        if norm_layer:
            out.add_module("normalization", norm_layer(num_features=out_filters))
        if nonlinearity:
            out.add_module("nonlinearity", nonlinearity)
        return out

    @staticmethod
    def resnet_block(in_filters, out_filters, size, stride, padding, bias, norm_layer,
                     nonlinearity):
        out = nn.Sequential()
        out.add_module("nonlinearity_0", nonlinearity)
        out.add_module("conv_0", nn.Conv2d(in_filters, out_filters, kernel_size=size,
                                           stride=stride, padding=padding, bias=bias))
        if norm_layer:
            out.add_module("normalization", norm_layer(num_features=out_filters))
        out.add_module("nonlinearity_1", nonlinearity)
        out.add_module("conv_1", nn.Conv2d(in_filters, out_filters, kernel_size=size,
                                           stride=stride, padding=padding, bias=bias))
        return out

    @staticmethod
    def upconv_layer_upsample_and_conv(in_filters, out_filters, bias, norm_layer, nonlinearity):
        parts = [UpsamplingLayer(), nn.Conv2d(in_filters, out_filters, 3, 1, 1, bias=False)]
        if norm_layer:
            parts.append(norm_layer(num_features=out_filters))
        if nonlinearity:
            parts.append(nonlinearity)
        return nn.Sequential(*parts)

    # ---- fused conv dispatch
    def _conv(self, x, conv, bn=None, act=None, residual=None, in_relu=False, coords=None):
        scale = shift = None
        if bn is not None:
            scale, shift = _bn_fold(bn)
        # evaluation: the packed-weight kernels (ops.PackedConvWeight), bf16 x 3 or exact f32 per
        # operator kind (EVAL_X3 / EVAL_DEFORM_X3 above)
        if coords is not None:
            assert conv.bias is None
            return ops.deform_conv3x3_x3(x, coords, _packed(conv, not EVAL_DEFORM_X3), scale, shift,
                                         act, residual, in_relu)
        exact = not EVAL_X3
        if not _packed_ok(conv, exact):
            if x.shape[1] != conv.weight.shape[1]:
                x = x[:, :conv.weight.shape[1]].contiguous()      # zero channels of _cat_in
            return ops.conv2d(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], scale,
                              shift, act, residual, in_relu)
        return ops.conv2d_x3(x, _packed(conv, exact), conv.bias, conv.stride[0], conv.padding[0],
                             scale, shift, act, residual, in_relu)

    def _check(self, x):
        if not x.is_cuda:
            raise RuntimeError("gfx950 generators need a device tensor (no CPU fallback)")

    # ---- training-mode building blocks (batch statistics; one kernel per op)
    #: how many times the running statistics take each batch.  The reference evaluates the
    #: generator twice per iteration on the same batch with unchanged weights (discriminator
    #: step, then generator step: trainers.py:88,102); a trainer that shares one forward between
    #: the two steps sets this to 2 so that the BatchNorm buffers end up identical.
    stat_updates = 1

    def _tconv(self, x, conv, plan=None, act=None):
        if plan is not None:
            return Fn.conv(x, conv.weight, None, 1, 1, act, plan)
        return Fn.conv(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], act)

    def _tbn(self, x, bn, act=None):
        return Fn.batch_norm_train(x, bn, act, self.stat_updates)

    def _tfinal(self, output, plan=None):
        if isinstance(self.conv_12, nn.Sequential):
            return self._tconv(output, self.conv_12[0], act="tanh")
        return self._tconv(output, self.conv_12)


class GeneratorJ(_GeneratorBase):
    """Stage-2 generator (models.py:24-192): plain convolutions."""
    RIC = False

    def forward(self, x):
        self._check(x)
        x = x.float().contiguous()
        if self.training:
            return self._forward_train(x)
        c0, c1, c2 = self.conv0, self.conv1, self.conv2
        output_0 = self._conv(x, c0.conv, c0.normalization, "leaky_relu")
        output_1 = self._conv(output_0, c1.conv, c1.normalization, "leaky_relu")
        output_2 = self._conv(output_1, c2.conv, c2.normalization, "leaky_relu")
        output = output_2
        for layer in self.resnets:
            tmp = self._conv(output, layer.conv_0, layer.normalization, "relu", in_relu=True)
            output = self._conv(tmp, layer.conv_1, residual=output)
        output = self._up(self.upconv2, torch.cat((output, output_2), 1))
        output = self._up(self.upconv1, torch.cat((output, output_1), 1))
        output = self._conv(_cat_in((output, output_0, x)), self.conv_11[0], act="relu")
        if self.append_smoothers:
            a = self.conv_11_a
            # conv -> ReLU -> BN -> conv -> ReLU  (BN comes AFTER the ReLU here: models.py:98-104)
            tmp = self._conv(output, a[0], act="relu")
            scale, shift = _bn_fold(a[2])
            tmp = tmp * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
            output = self._conv(tmp, a[3], act="relu")
        return self._final(output)

    def _up(self, seq, x):
        return self._conv(ops.upsample2_fwd(x), seq[1], seq[2], _act_name(seq[3]))

    def _forward_train(self, x):
        """models.py:113-129 with train-mode BatchNorm."""
        assert self.norm_layer is not None, "training path expects norm_layer='batch_norm'"
        c0, c1, c2 = self.conv0, self.conv1, self.conv2
        output_0 = self._tbn(self._tconv(x, c0.conv), c0.normalization, "leaky_relu")
        output_1 = self._tbn(self._tconv(output_0, c1.conv), c1.normalization, "leaky_relu")
        output_2 = self._tbn(self._tconv(output_1, c2.conv), c2.normalization, "leaky_relu")
        output = output_2
        for layer in self.resnets:
            tmp = self._tconv(Fn.activation(output, "relu"), layer.conv_0)
            tmp = self._tbn(tmp, layer.normalization, "relu")
            output = self._tconv(tmp, layer.conv_1) + output
        for seq, skip in ((self.upconv2, output_2), (self.upconv1, output_1)):
            tmp = Fn.upsample2(torch.cat((output, skip), 1))
            output = self._tbn(self._tconv(tmp, seq[1]), seq[2], _act_name(seq[3]))
        output = self._tconv(torch.cat((output, output_0, x), 1), self.conv_11[0], act="relu")
        if self.append_smoothers:
            a = self.conv_11_a           # conv -> ReLU -> BN -> conv -> ReLU (models.py:98-104)
            tmp = self._tbn(self._tconv(output, a[0], act="relu"), a[2])
            output = self._tconv(tmp, a[3], act="relu")
        return self._tfinal(output)

    def _final(self, output):
        if isinstance(self.conv_12, nn.Sequential):
            return self._conv(output, self.conv_12[0], act="tanh")
        return self._conv(output, self.conv_12)


class GeneratorJ_RIC(_GeneratorBase):
    """Stage-1 generator (models.py:200-356): every 3x3 convolution is a deformable convolution
    with the fixed rotation-invariant offsets of generate_coordinates."""
    RIC = True

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.coords_0 = self.coords_1 = self.coords_2 = None
        self.current_x_shape = None

    def _forward_train(self, x):
        """models.py:293-356 with train-mode BatchNorm.  The offset maps depend on the
        resolution only, so each has one cached DeformPlan (sampling table + its transpose)."""
        assert self.norm_layer is not None, "training path expects norm_layer='batch_norm'"
        p0, p1, p2 = (ops.deform_plan(c[0]) for c in (self.coords_0, self.coords_1, self.coords_2))
        c0, c1, c2 = self.conv0, self.conv1, self.conv2
        output_0 = self._tbn(self._tconv(x, c0.conv, p0), c0.normalization, "leaky_relu")
        tmp = self._tconv(Fn.maxpool2(output_0), c1.conv, p1)
        output_1 = self._tbn(tmp, c1.normalization, "leaky_relu")
        tmp = self._tconv(Fn.maxpool2(output_1), c2.conv, p2)
        output_2 = self._tbn(tmp, c2.normalization, "leaky_relu")
        output = output_2
        for layer in self.resnets:
            tmp = self._tconv(Fn.activation(output, "relu"), layer.conv_0, p2)
            tmp = self._tbn(tmp, layer.normalization, "relu")
            output = self._tconv(tmp, layer.conv_1, p2) + output
        tmp = Fn.upsample2(torch.cat((output, output_2), 1))
        output = self._tbn(self._tconv(tmp, self.upconv2[1], p1), self.upconv2[2], "relu")
        tmp = Fn.upsample2(torch.cat((output, output_1), 1))
        output = self._tbn(self._tconv(tmp, self.upconv1[1], p0), self.upconv1[2], "relu")
        output = self._tconv(torch.cat((output, output_0, x), 1), self.conv_11[0], p0, "relu")
        if self.append_smoothers:
            a = self.conv_11_a
            # models.py:347-352: conv_11_a[0..2] run (their BatchNorm buffers move) but the
            # second convolution reads `output`, so they receive no gradient.
            with torch.no_grad():
                self._tbn(self._tconv(output, a[0], p0, "relu"), a[2])
            output = self._tconv(output, a[3], p0, "relu")
        return self._tfinal(output)

    def forward(self, x):
        self._check(x)
        x = x.float().contiguous()
        if x.shape != self.current_x_shape:
            self.current_x_shape = x.shape
            B, _, H, W = x.shape
            self.coords_0 = generate_coordinates(B, H, W, x.device)
            self.coords_1 = generate_coordinates(B, int(H / 2), int(W / 2), x.device)
            self.coords_2 = generate_coordinates(B, int(H / 4), int(W / 4), x.device)
        if self.training:
            return self._forward_train(x)
        k0, k1, k2 = self.coords_0[0], self.coords_1[0], self.coords_2[0]
        c0, c1, c2 = self.conv0, self.conv1, self.conv2
        output_0 = self._conv(x, c0.conv, c0.normalization, "leaky_relu", coords=k0)
        output_1 = self._conv(self.maxpool(output_0), c1.conv, c1.normalization, "leaky_relu",
                              coords=k1)
        output_2 = self._conv(self.maxpool(output_1), c2.conv, c2.normalization, "leaky_relu",
                              coords=k2)
        output = output_2
        for layer in self.resnets:
            tmp = self._conv(output, layer.conv_0, layer.normalization, "relu", in_relu=True,
                             coords=k2)
            output = self._conv(tmp, layer.conv_1, residual=output, coords=k2)
        tmp = ops.upsample2_fwd(torch.cat((output, output_2), 1))
        output = self._conv(tmp, self.upconv2[1], self.upconv2[2], "relu", coords=k1)
        tmp = ops.upsample2_fwd(torch.cat((output, output_1), 1))
        output = self._conv(tmp, self.upconv1[1], self.upconv1[2], "relu", coords=k0)
        output = self._conv(_cat_in((output, output_0, x)), self.conv_11[0], act="relu",
                            coords=k0)
        if self.append_smoothers:
            # models.py:347-352: the second smoother convolution takes `output` (the conv_11
            # result), not the first smoother's result, so conv_11_a[0..2] are dead at
            # inference and are skipped here; the value is ReLU(deform(output, conv_11_a[3])).
            output = self._conv(output, self.conv_11_a[3], act="relu", coords=k0)
        if isinstance(self.conv_12, nn.Sequential):
            return self._conv(output, self.conv_12[0], act="tanh")
        return self._conv(output, self.conv_12)


def build_model(model_type, args, device=None):
    """training/trainers.py:33-35 build_model(type, args, device)."""
    cls = {"GeneratorJ": GeneratorJ, "GeneratorJ_RIC": GeneratorJ_RIC}[model_type]
    model = cls(**args)
    return model.to(device) if device else model
