"""Side-view mattes of mv.py (2_charactor_reconstructor/mv.py:18,113-150): `remove_background(session,
image)` on the predicted colour (or normal) image of the four side views, where `session` is an
onnxruntime session of the IS-Net "DIS" matting network (`dis_pretrained/isnet_dis.onnx`).

What the reference itself contains is the function around the session — normalisation
`(x / 255 - 0.5) / 1.0`, CHW + batch axis, `ort_outs[0][0][0]`, clip to [0, 1], `* 255` to uint8 —
restated here as `remove_background` with the same signature, pinned to the reference's own function
(run with a stub session) by tests/golden/matting_reference.npz.

The network behind the session is third-party (xuebinqin/DIS `isnet.py`, exported to ONNX by the
reference's authors; neither the weights nor onnxruntime are in the snapshot or in this image):
`ISNetDIS` restates its architecture with the DIS repository's parameter names, so that the published
`isnet-general-use.pth` state_dict loads unchanged (`load_isnet(path)`); unpinned leaf, random
weights otherwise.  `IsnetSession` gives it the two session methods the reference calls
(`get_inputs()[0].name`, `run(None, {name: array})`), with every convolution on the library's HIP
kernel (`dsu_conv2d_fwd`, batch-norm folded into its epilogue; a dilated 3x3 convolution is d x d
ordinary ones on the sub-lattices x[i::d, j::d]).  `entry/mv.py --matting isnet` uses it for
`write_mv_outputs(matting_fn=...)`; the default stays the filled-silhouette stand-in
(`entry.data.side_mask_from_prediction`), since random weights do not segment anything.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from PIL import Image

from .. import ops


# ------------------------------------------------------------------ mv.py:129-150
def normalize(image, mean, std):
    """mv.py:129-131."""
    return (image / 255.0 - mean) / std


def remove_background(session, image_pil):
    """mv.py:134-150: PIL RGB image -> 'L' matte through `session` (anything with onnxruntime's
    `get_inputs()` / `run()`)."""
    im = np.array(image_pil, dtype=np.float32)
    im = normalize(im, mean=[0.5, 0.5, 0.5], std=[1.0, 1.0, 1.0])
    im = np.transpose(im, (2, 0, 1))
    im = np.expand_dims(im, axis=0)
    im = im.astype(np.float32)
    ort_inputs = {session.get_inputs()[0].name: im}
    ort_outs = session.run(None, ort_inputs)
    result = ort_outs[0][0][0]
    result = np.clip(result, 0, 1)
    return Image.fromarray((result * 255).astype(np.uint8))


# ------------------------------------------------------------------ IS-Net (DIS), restated
class REBNCONV(nn.Module):
    """conv3x3 (dilation d, padding d) + BatchNorm + ReLU; DIS names conv_s1 / bn_s1."""

    def __init__(self, in_ch=3, out_ch=3, dirate=1, stride=1):
        super().__init__()
        self.conv_s1 = nn.Conv2d(in_ch, out_ch, 3, padding=dirate, dilation=dirate, stride=stride)
        self.bn_s1 = nn.BatchNorm2d(out_ch)
        self.relu_s1 = nn.ReLU(inplace=True)

    def forward(self, x):
        if x.is_cuda and not self.training:
            return _rebnconv_hip(self, x)
        return self.relu_s1(self.bn_s1(self.conv_s1(x)))


# Convolution arithmetic on the device.  The reference runs the network as an f32 ONNX session
# (mv.py:17-18): False (default) = exact-f32 products on the f32 MFMA (dsu_conv2d_fwd_f32p); True =
# bf16 x 3 products on the bf16 MFMA (dsu_conv2d_fwd_x3, ~2^-16 relative per product: far inside
# what a matte thresholded at 127 / 255 needs, but not the reference's arithmetic — an A/B switch
# for tools and tests).
EVAL_X3 = False


def _packed(conv, exact):
    """The convolution's weight in the evaluation kernels' packed layout, cached on the module."""
    w = conv.weight
    ver = (w._version, w.data_ptr(), w.device, bool(exact))
    cache = getattr(conv, "_dsu_pack", None)
    if cache is None or cache[0] != ver:
        conv._dsu_pack = cache = (ver, ops.PackedConvWeight(w, exact=exact))
    return cache[1]


def _conv(x, conv, bias, stride, padding, scale=None, shift=None, act=None):
    return ops.conv2d_x3(x, _packed(conv, not EVAL_X3), bias, stride, padding, scale, shift, act)


def _fold(conv, bn):
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    if conv.bias is not None:
        shift = shift + scale * conv.bias
    return scale.float().contiguous(), shift.float().contiguous()


def _rebnconv_hip(m, x):
    """relu(bn(conv(x))) on dsu_conv2d_fwd.  Dilation d (padding d, stride 1): output pixel (y, x)
    reads x[y + d (i - 1), x + d (j - 1)], i.e. only pixels of its own residue class modulo d —
    an ordinary padding-1 convolution on each of the d x d sub-lattices."""
    conv = m.conv_s1
    d, s = conv.dilation[0], conv.stride[0]
    scale, shift = _fold(conv, m.bn_s1)
    x = x.float()
    if d == 1:
        return _conv(x.contiguous(), conv, None, s, 1, scale, shift, "relu")
    assert s == 1
    B, _, H, W = x.shape
    out = torch.empty((B, conv.weight.shape[0], H, W), dtype=torch.float32, device=x.device)
    for i in range(d):
        for j in range(d):
            sub = x[:, :, i::d, j::d]
            if sub.numel():
                out[:, :, i::d, j::d] = _conv(sub.contiguous(), conv, None, 1, 1, scale, shift, "relu")
    return out


def _conv_hip(conv, x, stride=1):
    """plain nn.Conv2d 3x3 (conv_in, side outputs) on the HIP kernel when on the device."""
    if x.is_cuda:
        return _conv(x.float().contiguous(), conv,
                     None if conv.bias is None else conv.bias.float().contiguous(), stride, conv.padding[0])
    return conv(x)


def _upsample_like(src, tar):
    return F.interpolate(src, size=tar.shape[2:], mode="bilinear", align_corners=False)


class RSU(nn.Module):
    """Residual U-block of height `depth` (RSU7 ... RSU4): `depth - 1` encoder convolutions with
    2x2 ceil-mode max pooling between them, one dilated (2) bottom convolution, mirrored decoder on
    the concatenations, residual on the input convolution."""

    def __init__(self, depth, in_ch, mid_ch, out_ch):
        super().__init__()
        self.depth = depth
        self.rebnconvin = REBNCONV(in_ch, out_ch, 1)
        self.rebnconv1 = REBNCONV(out_ch, mid_ch, 1)
        for k in range(2, depth):
            setattr(self, f"rebnconv{k}", REBNCONV(mid_ch, mid_ch, 1))
        setattr(self, f"rebnconv{depth}", REBNCONV(mid_ch, mid_ch, 2))
        for k in range(depth - 1, 1, -1):
            setattr(self, f"rebnconv{k}d", REBNCONV(mid_ch * 2, mid_ch, 1))
        self.rebnconv1d = REBNCONV(mid_ch * 2, out_ch, 1)

    def forward(self, x):
        hxin = self.rebnconvin(x)
        enc = [self.rebnconv1(hxin)]
        for k in range(2, self.depth):
            enc.append(getattr(self, f"rebnconv{k}")(F.max_pool2d(enc[-1], 2, stride=2, ceil_mode=True)))
        hx = getattr(self, f"rebnconv{self.depth}")(enc[-1])
        for k in range(self.depth - 1, 0, -1):
            hx = getattr(self, f"rebnconv{k}d")(torch.cat((hx, enc[k - 1]), 1))
            if k > 1:
                hx = _upsample_like(hx, enc[k - 2])
        return hx + hxin


class RSU4F(nn.Module):
    """The dilated-only block of the two coarsest stages (dilations 1, 2, 4, 8; no pooling)."""

    def __init__(self, in_ch, mid_ch, out_ch):
        super().__init__()
        self.rebnconvin = REBNCONV(in_ch, out_ch, 1)
        self.rebnconv1 = REBNCONV(out_ch, mid_ch, 1)
        self.rebnconv2 = REBNCONV(mid_ch, mid_ch, 2)
        self.rebnconv3 = REBNCONV(mid_ch, mid_ch, 4)
        self.rebnconv4 = REBNCONV(mid_ch, mid_ch, 8)
        self.rebnconv3d = REBNCONV(mid_ch * 2, mid_ch, 4)
        self.rebnconv2d = REBNCONV(mid_ch * 2, mid_ch, 2)
        self.rebnconv1d = REBNCONV(mid_ch * 2, out_ch, 1)

    def forward(self, x):
        hxin = self.rebnconvin(x)
        hx1 = self.rebnconv1(hxin)
        hx2 = self.rebnconv2(hx1)
        hx3 = self.rebnconv3(hx2)
        hx4 = self.rebnconv4(hx3)
        hx3d = self.rebnconv3d(torch.cat((hx4, hx3), 1))
        hx2d = self.rebnconv2d(torch.cat((hx3d, hx2), 1))
        hx1d = self.rebnconv1d(torch.cat((hx2d, hx1), 1))
        return hx1d + hxin


class ISNetDIS(nn.Module):
    """xuebinqin/DIS IS-Net: stride-2 input convolution, six RSU stages down, five up, one 3x3 side
    output per decoder stage; the exported model's first output is sigmoid(side1) at the input size."""

    def __init__(self, in_ch=3, out_ch=1):
        super().__init__()
        self.conv_in = nn.Conv2d(in_ch, 64, 3, stride=2, padding=1)
        self.stage1 = RSU(7, 64, 32, 64)
        self.stage2 = RSU(6, 64, 32, 128)
        self.stage3 = RSU(5, 128, 64, 256)
        self.stage4 = RSU(4, 256, 128, 512)
        self.stage5 = RSU4F(512, 256, 512)
        self.stage6 = RSU4F(512, 256, 512)
        self.stage5d = RSU4F(1024, 256, 512)
        self.stage4d = RSU(4, 1024, 128, 256)
        self.stage3d = RSU(5, 512, 64, 128)
        self.stage2d = RSU(6, 256, 32, 64)
        self.stage1d = RSU(7, 128, 16, 64)
        self.side1 = nn.Conv2d(64, out_ch, 3, padding=1)
        self.side2 = nn.Conv2d(64, out_ch, 3, padding=1)
        self.side3 = nn.Conv2d(128, out_ch, 3, padding=1)
        self.side4 = nn.Conv2d(256, out_ch, 3, padding=1)
        self.side5 = nn.Conv2d(512, out_ch, 3, padding=1)
        self.side6 = nn.Conv2d(512, out_ch, 3, padding=1)

    def forward(self, x):
        """(B, 3, H, W) normalised image -> sigmoid(side1) upsampled to (B, 1, H, W)."""
        pool = lambda t: F.max_pool2d(t, 2, stride=2, ceil_mode=True)
        hxin = _conv_hip(self.conv_in, x, 2)
        hx1 = self.stage1(hxin)
        hx2 = self.stage2(pool(hx1))
        hx3 = self.stage3(pool(hx2))
        hx4 = self.stage4(pool(hx3))
        hx5 = self.stage5(pool(hx4))
        hx6 = self.stage6(pool(hx5))
        hx5d = self.stage5d(torch.cat((_upsample_like(hx6, hx5), hx5), 1))
        hx4d = self.stage4d(torch.cat((_upsample_like(hx5d, hx4), hx4), 1))
        hx3d = self.stage3d(torch.cat((_upsample_like(hx4d, hx3), hx3), 1))
        hx2d = self.stage2d(torch.cat((_upsample_like(hx3d, hx2), hx2), 1))
        hx1d = self.stage1d(torch.cat((_upsample_like(hx2d, hx1), hx1), 1))
        d1 = _upsample_like(_conv_hip(self.side1, hx1d), x)
        return torch.sigmoid(d1)


def load_isnet(path=None, device="cuda", seed=0):
    """ISNetDIS in eval mode; `path`: a DIS state_dict (`isnet-general-use.pth`; keys of the
    unused training-time modules are ignored), None: seeded random weights."""
    if path is None:
        # seeded random weights without touching the caller's global RNG stream
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(seed)
            net = ISNetDIS()
    else:
        net = ISNetDIS()
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
        missing, unexpected = net.load_state_dict(sd, strict=False)
        if missing:
            raise KeyError(f"{path}: missing IS-Net parameters, e.g. {missing[:4]}")
    return net.eval().to(device)


class _Input:
    name = "input_image"


class IsnetSession:
    """The part of onnxruntime.InferenceSession that mv.py uses, on an ISNetDIS module."""

    def __init__(self, net, device=None):
        self.net = net
        self.device = torch.device(device) if device is not None else next(net.parameters()).device

    def get_inputs(self):
        return [_Input()]

    @torch.no_grad()
    def run(self, output_names, feed):
        x = torch.from_numpy(np.ascontiguousarray(feed[_Input.name])).to(self.device)
        return [self.net(x).float().cpu().numpy()]


def matting_fn(session):
    """`matting_fn` of entry.data.write_mv_outputs: PIL image -> 'L' matte (mv.py:120-122)."""
    return lambda image_pil: remove_background(session, image_pil.convert("RGB"))
