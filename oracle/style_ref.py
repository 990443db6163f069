"""CPU oracle for the style-translator path.  TEST INFRASTRUCTURE ONLY.

  * deform_conv2d: restates torchvision==0.15.1 ops/deform_conv2d (un-vendored,
    README.md:29; call sites 3_style_translator/training/models.py:302-351): bilinear
    im2col (zero outside (-1,H)x(-1,W), per-corner validity) followed by a GEMM.
    PARITY UNPINNED for the op itself (no torchvision here, no reference tests); KATs:
    zero offsets == F.conv2d, integer offsets == shifted conv.
  * generate_coordinates: the reference's own function (models.py:551-604) is imported by
    tests/golden/make_style_golden.py; this restatement is pinned by that fixture.
"""
import math

import torch
import torch.nn.functional as F


def deform_conv2d(inp, offset, weight, padding=(1, 1)):
    """inp (B,C,H,W), offset (B,2*kh*kw,H,W) [channel 2k = dy, 2k+1 = dx], weight (O,C,kh,kw);
    stride 1, dilation 1, one offset group, no mask/bias.  float64 internally."""
    B, C, H, W = inp.shape
    O, _, kh, kw = weight.shape
    ph, pw = padding
    dt = torch.float64
    x = inp.to(dt)
    off = offset.to(dt)
    ys = torch.arange(H, dtype=dt).view(1, H, 1)
    xs = torch.arange(W, dtype=dt).view(1, 1, W)
    cols = []
    for k in range(kh * kw):
        i, j = k // kw, k % kw
        h = ys - ph + i + off[:, 2 * k]       # (B,H,W)
        w = xs - pw + j + off[:, 2 * k + 1]
        inside = (h > -1) & (w > -1) & (h < H) & (w < W)
        h0 = torch.floor(h); w0 = torch.floor(w)
        lh = h - h0; lw = w - w0
        h0 = h0.long(); w0 = w0.long(); h1 = h0 + 1; w1 = w0 + 1

        def corner(hi, wi, wt):
            valid = inside & (hi >= 0) & (hi <= H - 1) & (wi >= 0) & (wi <= W - 1)
            idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).view(B, 1, H * W).expand(B, C, H * W)
            v = torch.gather(x.view(B, C, H * W), 2, idx).view(B, C, H, W)
            return v * (wt * valid).unsqueeze(1)

        val = corner(h0, w0, (1 - lh) * (1 - lw)) + corner(h0, w1, (1 - lh) * lw) + \
            corner(h1, w0, lh * (1 - lw)) + corner(h1, w1, lh * lw)
        cols.append(val)
    col = torch.stack(cols, 2)                # (B,C,K,H,W)
    out = torch.einsum("ock,bckhw->bohw", weight.to(dt).reshape(O, C, kh * kw), col)
    return out


def generate_coordinates(H, W):
    """(18,H,W) f32 offset map of GeneratorJ_RIC, same torch op sequence as models.py:551-604."""
    coords = torch.zeros(H, W, 18)
    center = torch.zeros(2)
    p = torch.zeros(3); p[1] = H; p[2] = W
    center[0] = torch.sub(torch.div(p[1], 2.0), 0.5)
    center[1] = torch.sub(torch.div(p[2], 2.0), 0.5)
    gx, gy = torch.meshgrid(torch.arange(0, p[1]), torch.arange(0, p[2]), indexing="ij")
    dx = torch.sub(gx, center[0]); dy = torch.sub(gy, center[1])
    PI = torch.mul(torch.Tensor([math.pi]), 2.0)
    theta = torch.atan2(dy, dx) % PI[0]
    theta = torch.round(10000. * theta) / 10000.
    base = [(1., 1.), (1., 0.), (1., -1.), (0., 1.), None, (0., -1.), (-1., 1.), (-1., 0.), (-1., -1.)]
    for k in range(9):
        if k == 4:
            continue
        m = float(k if k < 4 else k - 1)
        ang = torch.add(theta, torch.mul(torch.div(PI[0], 8.0), m))
        coords[:, :, 2 * k] = torch.add(torch.cos(ang), base[k][0])
        coords[:, :, 2 * k + 1] = torch.add(torch.sin(ang), base[k][1])
    return coords.permute(2, 0, 1).contiguous()


def conv_bn_act(x, weight, bias, stride, padding, bn=None, act=None, residual=None):
    """nn.Conv2d -> eval BatchNorm2d -> activation (+ residual), float64."""
    dt = torch.float64
    y = F.conv2d(x.to(dt), weight.to(dt), None if bias is None else bias.to(dt), stride, padding)
    if bn is not None:
        g, b, mean, var, eps = bn
        y = (y - mean.to(dt).view(1, -1, 1, 1)) / torch.sqrt(var.to(dt).view(1, -1, 1, 1) + eps) \
            * g.to(dt).view(1, -1, 1, 1) + b.to(dt).view(1, -1, 1, 1)
    if act == "relu":
        y = F.relu(y)
    elif act == "leaky_relu":
        y = F.leaky_relu(y, 0.2)
    elif act == "tanh":
        y = torch.tanh(y)
    if residual is not None:
        y = y + residual.to(dt)
    return y


def seeded_state_dict(template, seed):
    """Deterministic parameters for a generator from a SEED (fixtures store the seed, not 13 MB of
    weights): every tensor of `template` (a state_dict, any module tree) is filled from its own
    generator keyed by crc32(name) + seed — convolution weights N(0, 1 / fan_in) (the last
    1x1 convolution N(0, 3 / fan_in): output spread ~0.3, tanh not saturated), BatchNorm
    weight/var in [0.5, 1.5], bias/mean N(0, 0.1^2), integer buffers kept."""
    import zlib
    out = {}
    for name in sorted(template):
        t = template[name]
        if not t.dtype.is_floating_point:
            out[name] = t.clone()
            continue
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))
        if t.dim() > 1:
            fan_in = t[0].numel()
            gain = 3.0 if (t.dim() == 4 and t.shape[0] == 3) else 1.0     # the RGB head
            v = torch.randn(t.shape, generator=g) * (gain / fan_in) ** 0.5
        elif name.endswith("running_var") or name.endswith("weight"):
            v = torch.rand(t.shape, generator=g) + 0.5
        else:
            v = torch.randn(t.shape, generator=g) * 0.1
        out[name] = v.to(t.dtype)
    return out


# shared by tests/golden/make_style_fullsize_golden.py and the GPU test that reads its fixture
FULLSIZE_ARGS = dict(use_bias=False, tanh=True, append_smoothers=True, resnet_blocks=7,
                     filters=[32, 64, 128, 128, 128, 64], input_channels=6)   # config_stage{1,2}.yaml


def fullsize_frame(seed):
    """(1,6,512,512) in [-1,1]: smooth random field + 5 % noise (a frame-like input from a seed)."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, 6, 64, 64, generator=g) * 2 - 1
    x = F.interpolate(low, size=(512, 512), mode="bilinear", align_corners=False)
    return (x + 0.05 * torch.randn(1, 6, 512, 512, generator=g)).clamp(-1, 1)
