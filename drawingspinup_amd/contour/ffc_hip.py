"""The FFC-ResNet generator (contour/ffc.py = saicinpainting/training/modules/ffc.py:48-366) on the
library's own exact-f32 MFMA convolution kernel (csrc/style_conv.hip, `dsu_conv2d_fwd`): every
convolution, the 1x1 layers of the spectral transform, the transposed convolutions of the decoder
and the dense DFT products of the FourierUnit run through `ops.conv2d` — no MIOpen / rocBLAS call
is left on the contour path (a fresh box paid ~9 s of MIOpen solver search in its first forward).

  * reflect-padded convolutions (`padding_mode='reflect'`, ffc.py:176-190): the padding is a copy
    (torch), the convolution runs unpadded on it;
  * FFC.forward (ffc.py:213-224): `convl2l(x_l) + convg2l(x_g)` is ONE convolution over the
    concatenated input channels with the concatenated weights; eval BatchNorm + ReLU of
    FFC_BN_ACT are the kernel's epilogue; `convl2g(x_l)` is handed to the spectral branch's last
    1x1 convolution as its residual;
  * ConvTranspose2d(k=3, s=2, p=1, output_padding=1) (ffc.py:345-349) = stride-1 convolution of
    the zero-dilated input with the flipped / transposed weights;
  * rfft2 / irfft2 (`norm='ortho'`): products with the cos / sin matrices as 1x1 convolutions whose
    "channel" axis is the transformed axis (H first, one transposition, then W; the pointwise
    layers between the two transforms do not care that their spatial layout is (W/2+1, H)).
Inference only (eval BatchNorm folded).  Parity: tests/test_contour_host.py (fixture made by the
reference's own modules), same tolerance as the torch-operator form.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from . import ffc as M


def _fold(bn):
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous()


def _conv(x, w, stride=1, pad=0, bn=None, act=None, residual=None, bias=None):
    if pad:
        x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
    sc = sh = None
    if bn is not None:
        sc, sh = _fold(bn)
        if bias is not None:
            sh, bias = sh + sc * bias, None
    return ops.conv2d(x.contiguous(), w.contiguous(), bias, stride, 0, sc, sh, act, residual)


def _left_mul(a, x):
    """a (p, q) applied along axis -2 of x (n, q, k) -> (n, p, k), as a 1x1 convolution whose
    channels are that axis."""
    n, q, k = x.shape
    return ops.conv2d(x.reshape(n, q, 1, k).contiguous(), a.reshape(a.shape[0], q, 1, 1).contiguous()
                      ).reshape(n, a.shape[0], k)


def _mats(h, w, device):
    key = ("hip", h, w, str(device))
    m = M._DFT_CACHE.get(key)
    if m is None:
        ch, sh = M._dft_mats(h, False, device, torch.float32)
        cw, sw = M._dft_mats(w, True, device, torch.float32)
        k = w // 2 + 1
        mult = torch.full((k,), 2.0, device=device)
        mult[0] = 1.0
        if w % 2 == 0:
            mult[-1] = 1.0
        fwd_h = torch.cat([ch.t(), sh.t()], 0)                       # (2h, h): [C x ; S x]
        fwd_w = torch.cat([cw.t(), sw.t()], 0)                       # (2k, w)
        cm, sm = (cw * mult).contiguous(), (sw * mult).contiguous()  # (w, k)
        inv_w = torch.cat([torch.cat([cm, -sm], 1), torch.cat([sm, cm], 1)], 0)   # (2w, 2k)
        inv_h = torch.cat([ch, -sh], 1)                              # (h, 2h): cos Re U - sin Im U
        m = tuple(t.contiguous() for t in (fwd_h, fwd_w, inv_w, inv_h))
        from .._lib import publish_sync
        publish_sync(device)
        M._DFT_CACHE[key] = m
    return m


def fourier_unit(fu, x):
    b, c, h, w = x.shape
    k = w // 2 + 1
    fwd_h, fwd_w, inv_w, inv_h = _mats(h, w, x.device)
    ab = _left_mul(fwd_h, x.reshape(b * c, h, w))                          # (bc, 2h, w): [A ; B]
    abt = ab.reshape(b * c * 2, h, w).transpose(1, 2).contiguous()        # (2bc, w, h)
    q = _left_mul(fwd_w, abt).reshape(b * c, 2, 2, k, h)                  # [blk A/B][cos/sin](k,h)
    re = q[:, 0, 0] - q[:, 1, 1]                                          # A cw - B sw
    im = -(q[:, 1, 0] + q[:, 0, 1])                                       # -(B cw + A sw)
    z = torch.stack((re, im), 1).reshape(b, 2 * c, k, h)                  # channel = c * 2 + {re, im}
    z = _conv(z, fu.conv_layer.weight, bn=fu.bn, act="relu")
    c2 = z.shape[1] // 2
    u = _left_mul(inv_w, z.reshape(b * c2, 2 * k, h))                     # (bc2, 2w, h): [Re U ; Im U]
    ut = u.reshape(b * c2 * 2, w, h).transpose(1, 2).contiguous()         # (2bc2, h, w)
    return _left_mul(inv_h, ut.reshape(b * c2, 2 * h, w)).reshape(b, c2, h, w)


def spectral_transform(st, x, residual=None):
    x = st.downsample(x)
    x = _conv(x, st.conv1[0].weight, bn=st.conv1[1], act="relu")
    return _conv(x + fourier_unit(st.fu, x), st.conv2.weight, residual=residual)


def ffc_bn_act(m, x_l, x_g):
    f = m.ffc
    has_g_in, ref = torch.is_tensor(x_g), None
    for conv in (f.convl2l, f.convl2g, f.convg2l):
        if isinstance(conv, nn.Conv2d):
            ref = conv
    stride, pad = ref.stride[0], ref.padding[0]
    relu = lambda a: "relu" if isinstance(a, nn.ReLU) else None
    out_l = out_g = 0
    if f.ratio_gout != 1:
        if has_g_in:
            out_l = _conv(torch.cat((x_l, x_g), 1), torch.cat((f.convl2l.weight, f.convg2l.weight), 1),
                          stride, pad, bn=m.bn_l, act=relu(m.act_l))
        else:
            out_l = _conv(x_l, f.convl2l.weight, stride, pad, bn=m.bn_l, act=relu(m.act_l))
    if f.ratio_gout != 0:
        pre = _conv(x_l, f.convl2g.weight, stride, pad)
        if has_g_in:
            pre = spectral_transform(f.convg2g, x_g, residual=pre.contiguous())
        sc, sh = _fold(m.bn_g)
        out_g = pre * sc[None, :, None, None] + sh[None, :, None, None]
        if isinstance(m.act_g, nn.ReLU):
            out_g = torch.relu_(out_g)
    return out_l, out_g


def conv_transpose_bn_relu(ct, bn, x):
    """ConvTranspose2d(k=3, stride=2, padding=1, output_padding=1) + BN + ReLU."""
    assert ct.kernel_size == (3, 3) and ct.stride == (2, 2) and ct.padding == (1, 1) \
        and ct.output_padding == (1, 1)
    b, c, h, w = x.shape
    z = torch.zeros(b, c, 2 * h + 2, 2 * w + 2, device=x.device, dtype=x.dtype)   # k-1-p = 1 before, 1 + output_padding after
    z[:, :, 1:2 * h:2, 1:2 * w:2] = x
    wt = ct.weight.permute(1, 0, 2, 3).flip(2, 3)
    return _conv(z, wt, bn=bn, act="relu", bias=ct.bias)


@torch.no_grad()
def generator_forward(gen, x):
    layers = list(gen.model)
    x_l, x_g, i = x.float(), 0, 0
    while i < len(layers):
        m = layers[i]
        if isinstance(m, nn.ReflectionPad2d):
            p = m.padding[0]
            x_l = F.pad(x_l, (p, p, p, p), mode="reflect")
        elif isinstance(m, M.FFC_BN_ACT):
            x_l, x_g = ffc_bn_act(m, x_l, x_g)
        elif isinstance(m, M.FFCResnetBlock):
            y_l, y_g = ffc_bn_act(m.conv1, x_l, x_g)
            y_l, y_g = ffc_bn_act(m.conv2, y_l, y_g)
            x_l, x_g = x_l + y_l, x_g + y_g
        elif isinstance(m, M.ConcatTupleLayer):
            x_l = torch.cat((x_l, x_g), 1) if torch.is_tensor(x_g) else x_l
        elif isinstance(m, nn.ConvTranspose2d):
            x_l = conv_transpose_bn_relu(m, layers[i + 1], x_l)
            assert isinstance(layers[i + 2], nn.ReLU)
            i += 2
        elif isinstance(m, nn.Conv2d):
            x_l = _conv(x_l, m.weight, m.stride[0], m.padding[0], bias=m.bias)
        elif isinstance(m, nn.Sigmoid):
            x_l = torch.sigmoid(x_l)
        elif isinstance(m, nn.Tanh):
            x_l = torch.tanh(x_l)
        else:
            raise NotImplementedError(type(m).__name__)
        i += 1
    return x_l
