"""FFC-ResNet generator (LaMa "fourier" inpainting net) for the contour-removal stage.

Restates 1_lama_contour_remover/saicinpainting/training/modules/ffc.py:48-366 (FourierUnit,
SpectralTransform, FFC, FFC_BN_ACT, FFCResnetBlock, FFCResNetGenerator) for the configuration the
reference runs (configs/prediction/lama-fourier.yaml: 4 -> 1 channels, ngf 64, 3 downsamplings,
9 blocks at global ratio 0.75, no LFU, sigmoid output) with the SAME module tree, so that the
reference's `*_generator.ckpt` state_dict loads unchanged (`model.<i>.ffc.convl2l.weight`, ...).

What differs from the reference implementation:
  * the 2-D real FFT / inverse FFT of the FourierUnit (ffc.py:86,104, `norm='ortho'`) are evaluated
    as dense DFT products — rows then columns, `(H,H)` / `(W, W/2+1)` twiddle matrices built once
    per size in float64 — i.e. GEMMs on the matrix pipe instead of an FFT library call; at the
    64x64 bottleneck of a 512x512 drawing the DFT is 4 small GEMMs per direction;
  * only the options the shipped configuration uses are implemented (no gating, no SE block, no
    spectral positional encoding, no learnable spatial transform, no LFU branch); asking for
    them raises.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

# configs/prediction/lama-fourier.yaml:4-22 (the `${...}` references resolved)
LAMA_FOURIER_GENERATOR = dict(
    input_nc=4, output_nc=1, ngf=64, n_downsampling=3, n_blocks=9, add_out_act="sigmoid",
    init_conv_kwargs=dict(ratio_gin=0, ratio_gout=0, enable_lfu=False),
    downsample_conv_kwargs=dict(ratio_gin=0, ratio_gout=0, enable_lfu=False),
    resnet_conv_kwargs=dict(ratio_gin=0.75, ratio_gout=0.75, enable_lfu=False))

# lazily built, process-wide, NOT locked: the package drives one GPU from one Python thread per process
# (one process per GPU); guard with a mutex before calling these paths from several threads
_DFT_CACHE = {}


def _dft_mats(n, half, device, dtype):
    """(cos, sin) of 2*pi*j*k/n for j < n, k < (n//2+1 if half else n), scaled 1/sqrt(n)."""
    key = (n, half, str(device), dtype)
    m = _DFT_CACHE.get(key)
    if m is None:
        k = n // 2 + 1 if half else n
        ang = 2.0 * math.pi * torch.outer(torch.arange(n, dtype=torch.float64),
                                          torch.arange(k, dtype=torch.float64)) / n
        s = 1.0 / math.sqrt(n)
        m = ((torch.cos(ang) * s).to(device=device, dtype=dtype),
             (torch.sin(ang) * s).to(device=device, dtype=dtype))
        from .._lib import publish_sync
        publish_sync(device)
        _DFT_CACHE[key] = m
    return m


def rfft2_ortho(x):
    """torch.fft.rfftn(x, dim=(-2,-1), norm='ortho') as (real, imag), each (..., H, W//2+1)."""
    h, w = x.shape[-2:]
    cw, sw = _dft_mats(w, True, x.device, x.dtype)          # e^{-i t} = cos t - i sin t
    re, im = x @ cw, -(x @ sw)                               # along W
    ch, sh = _dft_mats(h, False, x.device, x.dtype)
    cht, sht = ch.t(), sh.t()
    # along H: (C - iS) (re + i im) = (C re + S im) + i (C im - S re)
    return cht @ re + sht @ im, cht @ im - sht @ re


def irfft2_ortho(re, im, h, w):
    """torch.fft.irfftn(complex(re, im), s=(h, w), dim=(-2,-1), norm='ortho')."""
    ch, sh = _dft_mats(h, False, re.device, re.dtype)
    # inverse along H (full complex): (C + iS)(re + i im)
    a = ch @ re - sh @ im
    b = ch @ im + sh @ re
    # inverse along W from the half spectrum: x[n] = sum_k m_k (a_k cos - b_k sin), m = 1 for the
    # DC (and Nyquist, even w) bins and 2 otherwise; the imaginary parts of those two bins are
    # ignored, as the library's C2R transform does
    cw, sw = _dft_mats(w, True, re.device, re.dtype)
    mult = torch.full((w // 2 + 1,), 2.0, device=re.device, dtype=re.dtype)
    mult[0] = 1.0
    if w % 2 == 0:
        mult[-1] = 1.0
    return (a * mult) @ cw.t() - (b * mult) @ sw.t()


class FourierUnit(nn.Module):
    """ffc.py:48-112: rfft2 -> [re | im] channel interleave -> 1x1 conv + BN + ReLU -> irfft2."""

    def __init__(self, in_channels, out_channels, groups=1):
        super().__init__()
        self.conv_layer = nn.Conv2d(in_channels * 2, out_channels * 2, 1, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(out_channels * 2)

    def forward(self, x):
        b, c, h, w = x.shape
        re, im = rfft2_ortho(x)
        z = torch.stack((re, im), dim=2).reshape(b, 2 * c, h, w // 2 + 1)   # (c, 2) -> 2c, re first
        z = F.relu(self.bn(self.conv_layer(z)))
        z = z.view(b, -1, 2, h, w // 2 + 1)
        return irfft2_ortho(z[:, :, 0], z[:, :, 1], h, w)


class SpectralTransform(nn.Module):
    """ffc.py:115-162 without the LFU branch (enable_lfu: false in the shipped config)."""

    def __init__(self, in_channels, out_channels, stride=1, groups=1, enable_lfu=False):
        super().__init__()
        if enable_lfu:
            raise NotImplementedError("local Fourier unit (enable_lfu) is not used by lama-fourier.yaml")
        self.stride = stride
        self.downsample = nn.AvgPool2d(2, 2) if stride == 2 else nn.Identity()
        mid = out_channels // 2
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, mid, 1, groups=groups, bias=False),
                                   nn.BatchNorm2d(mid), nn.ReLU(inplace=True))
        self.fu = FourierUnit(mid, mid, groups)
        self.conv2 = nn.Conv2d(mid, out_channels, 1, groups=groups, bias=False)

    def forward(self, x):
        x = self.conv1(self.downsample(x))
        return self.conv2(x + self.fu(x))


def _maybe_conv(cin, cout, k, stride, padding, dilation, bias, padding_type):
    if cin == 0 or cout == 0:
        return nn.Identity()
    return nn.Conv2d(cin, cout, k, stride, padding, dilation, 1, bias, padding_mode=padding_type)


class FFC(nn.Module):
    """ffc.py:165-224: local/global channel split with the four cross paths."""

    def __init__(self, in_channels, out_channels, kernel_size, ratio_gin, ratio_gout, stride=1,
                 padding=0, dilation=1, bias=False, enable_lfu=False, padding_type="reflect"):
        super().__init__()
        assert stride in (1, 2)
        in_g = int(in_channels * ratio_gin)
        out_g = int(out_channels * ratio_gout)
        in_l, out_l = in_channels - in_g, out_channels - out_g
        self.ratio_gin, self.ratio_gout, self.global_in_num = ratio_gin, ratio_gout, in_g
        args = (kernel_size, stride, padding, dilation, bias, padding_type)
        self.convl2l = _maybe_conv(in_l, out_l, *args)
        self.convl2g = _maybe_conv(in_l, out_g, *args)
        self.convg2l = _maybe_conv(in_g, out_l, *args)
        self.convg2g = nn.Identity() if in_g == 0 or out_g == 0 else \
            SpectralTransform(in_g, out_g, stride, 1, enable_lfu)
        self.gate = nn.Identity()                       # gated=False in every shipped config

    def forward(self, x):
        x_l, x_g = x if isinstance(x, tuple) else (x, 0)
        out_l = out_g = 0
        if self.ratio_gout != 1:
            out_l = self.convl2l(x_l) + self.convg2l(x_g)
        if self.ratio_gout != 0:
            out_g = self.convl2g(x_l) + self.convg2g(x_g)
        return out_l, out_g


class FFC_BN_ACT(nn.Module):
    """ffc.py:227-254."""

    def __init__(self, in_channels, out_channels, kernel_size, ratio_gin, ratio_gout, stride=1,
                 padding=0, dilation=1, bias=False, activation=True, padding_type="reflect",
                 enable_lfu=False):
        super().__init__()
        self.ffc = FFC(in_channels, out_channels, kernel_size, ratio_gin, ratio_gout, stride,
                       padding, dilation, bias, enable_lfu, padding_type)
        g = int(out_channels * ratio_gout)
        self.bn_l = nn.Identity() if ratio_gout == 1 else nn.BatchNorm2d(out_channels - g)
        self.bn_g = nn.Identity() if ratio_gout == 0 else nn.BatchNorm2d(g)
        relu = lambda on: nn.ReLU(inplace=True) if on and activation else nn.Identity()
        self.act_l, self.act_g = relu(ratio_gout != 1), relu(ratio_gout != 0)

    def forward(self, x):
        x_l, x_g = self.ffc(x)
        return self.act_l(self.bn_l(x_l)), self.act_g(self.bn_g(x_g))


class FFCResnetBlock(nn.Module):
    """ffc.py:257-291 (not inline, no spatial transform wrapper)."""

    def __init__(self, dim, padding_type="reflect", dilation=1, **conv_kwargs):
        super().__init__()
        mk = lambda: FFC_BN_ACT(dim, dim, 3, padding=dilation, dilation=dilation,
                                padding_type=padding_type, **conv_kwargs)
        self.conv1, self.conv2 = mk(), mk()

    def forward(self, x):
        x_l, x_g = x if isinstance(x, tuple) else (x, 0)
        y_l, y_g = self.conv2(self.conv1((x_l, x_g)))
        return x_l + y_l, x_g + y_g


class ConcatTupleLayer(nn.Module):
    def forward(self, x):
        x_l, x_g = x
        return x_l if not torch.is_tensor(x_g) else torch.cat((x_l, x_g), dim=1)


class FFCResNetGenerator(nn.Module):
    """ffc.py:304-366: ReflectionPad + 7x7 FFC, 3 stride-2 FFCs, n_blocks FFC res-blocks, concat,
    3 x (ConvTranspose2d + BN + ReLU), ReflectionPad + 7x7 conv, output activation — in one
    nn.Sequential named `model`, index for index as in the reference."""

    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9,
                 padding_type="reflect", init_conv_kwargs=None, downsample_conv_kwargs=None,
                 resnet_conv_kwargs=None, add_out_act=True, max_features=1024, **unsupported):
        super().__init__()
        for k, v in unsupported.items():
            if v not in (None, False, {}, []):
                raise NotImplementedError(f"FFCResNetGenerator option {k}={v!r} is not implemented")
        init_kw = dict(init_conv_kwargs or {})
        down_kw = dict(downsample_conv_kwargs or {})
        res_kw = dict(resnet_conv_kwargs or {})
        layers = [nn.ReflectionPad2d(3), FFC_BN_ACT(input_nc, ngf, 7, padding=0, **init_kw)]
        for i in range(n_downsampling):
            kw = dict(down_kw)
            if i == n_downsampling - 1:              # hand the global share to the res-blocks
                kw["ratio_gout"] = res_kw.get("ratio_gin", 0)
            cin, cout = min(max_features, ngf * 2 ** i), min(max_features, ngf * 2 ** (i + 1))
            layers.append(FFC_BN_ACT(cin, cout, 3, stride=2, padding=1, **kw))
        feats = min(max_features, ngf * 2 ** n_downsampling)
        layers += [FFCResnetBlock(feats, padding_type, **res_kw) for _ in range(n_blocks)]
        layers.append(ConcatTupleLayer())
        for i in range(n_downsampling):
            m = 2 ** (n_downsampling - i)
            cin, cout = min(max_features, ngf * m), min(max_features, int(ngf * m / 2))
            layers += [nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1),
                       nn.BatchNorm2d(cout), nn.ReLU(True)]
        layers += [nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, 7, padding=0)]
        if add_out_act:
            kind = "tanh" if add_out_act is True else add_out_act
            layers.append({"tanh": nn.Tanh, "sigmoid": nn.Sigmoid}[kind]())
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        # on the device, in eval mode: the library's own convolution kernel for every layer
        # (contour/ffc_hip.py).  The torch operators below are the host form BASELINE config 1 names
        # ("predict.py on CPU PyTorch") and the training-mode graph.
        if x.is_cuda and not self.training:
            from . import ffc_hip
            return ffc_hip.generator_forward(self, x)
        return self.model(x)


def make_generator(kind="ffc_resnet", **kwargs):
    """saicinpainting/training/modules/__init__.py:7-19 for the kind the reference predicts with."""
    if kind != "ffc_resnet":
        raise ValueError(f"generator kind {kind!r}: only ffc_resnet is on this path")
    return FFCResNetGenerator(**kwargs)
