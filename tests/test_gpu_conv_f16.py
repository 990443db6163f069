"""HIP f16 NHWC implicit-GEMM convolution vs torch CPU f32 convolution on the same f16 data."""
import pytest
import torch
import torch.nn.functional as F

from drawingspinup_amd import ops

gpu = pytest.mark.gpu


def _r(shape, seed, s=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * s).half()


@gpu
@pytest.mark.parametrize("B,C,O,H,W,k,stride,pad,up", [
    (2, 320, 320, 16, 16, 3, 1, 1, False),      # ResnetBlock2D conv
    (2, 640, 320, 8, 8, 3, 1, 1, False),
    (1, 8, 320, 32, 32, 3, 1, 1, False),        # conv_in (C=8)
    (2, 320, 4, 16, 16, 3, 1, 1, False),        # conv_out (O=4)
    (2, 320, 320, 16, 16, 3, 2, 1, False),      # Downsample2D
    (2, 640, 640, 8, 8, 3, 1, 1, True),         # Upsample2D (nearest x2 folded in)
    (3, 320, 640, 8, 8, 1, 1, 0, False),        # conv_shortcut / proj_in (1x1)
    (1, 1920, 1280, 4, 4, 3, 1, 1, False),      # widest up-block input
    (2, 72, 136, 9, 7, 3, 1, 1, False),         # ragged sizes
])
def test_conv_f16(dev, B, C, O, H, W, k, stride, pad, up):
    x = _r((B, C, H, W), 1)
    w = _r((O, C, k, k), 2, (C * k * k) ** -0.5)
    b = _r((O,), 3, 0.1)
    xin = F.interpolate(x.float(), scale_factor=2, mode="nearest") if up else x.float()
    ref = F.conv2d(xin, w.float(), b.float(), stride, pad)
    got = ops.conv2d_nhwc_f16(x.permute(0, 2, 3, 1).contiguous().to(dev), ops.conv_weight_okc(w).to(dev),
                              b.to(dev), k, stride, pad, up)
    torch.testing.assert_close(got.cpu().float().permute(0, 3, 1, 2), ref, rtol=2e-3, atol=2e-3)


@gpu
def test_conv_f16_fused_epilogue(dev):
    B, C, O, H, W = 2, 320, 640, 8, 8
    x, w, b = _r((B, C, H, W), 4), _r((O, C, 3, 3), 5, (C * 9) ** -0.5), _r((O,), 6, 0.1)
    tv, res = _r((B, O), 7, 0.5), _r((B, O, H, W), 8)
    ref = F.conv2d(x.float(), w.float(), b.float(), 1, 1) + tv.float()[:, :, None, None] + res.float()
    got = ops.conv2d_nhwc_f16(x.permute(0, 2, 3, 1).contiguous().to(dev), ops.conv_weight_okc(w).to(dev),
                              b.to(dev), 3, 1, 1, False, tv.to(dev),
                              res.permute(0, 2, 3, 1).contiguous().to(dev))
    torch.testing.assert_close(got.cpu().float().permute(0, 3, 1, 2), ref, rtol=3e-3, atol=3e-3)


@gpu
def test_conv_f16_identity_kat(dev):
    """1x1 identity weight returns the input exactly (asymmetric data: catches transposes)."""
    C = 64
    x = torch.arange(2 * 5 * 7 * C, dtype=torch.float32).view(2, 5, 7, C).remainder(97).half()
    w = torch.eye(C).view(C, C, 1, 1).half()
    got = ops.conv2d_nhwc_f16(x.to(dev), ops.conv_weight_okc(w).to(dev), None, 1, 1, 0)
    assert torch.equal(got.cpu(), x)


@gpu
@pytest.mark.parametrize("split", [1, 2, 7, 16])
def test_conv_f16_split_k_matches_plain(dev, split):
    """Split-K (f32 partials + reduce/epilogue kernel) against the single-pass kernel and torch,
    on a 4x4-level UNet shape with the full fused epilogue and on an O that is not a multiple of 4."""
    B, C, O, H, W = 3, 1280, 640, 4, 4
    x, w, b = _r((B, C, H, W), 14), _r((O, C, 3, 3), 15, (C * 9) ** -0.5), _r((O,), 16, 0.1)
    tv, res = _r((B, O), 17, 0.5), _r((B, O, H, W), 18)
    ref = F.conv2d(x.float(), w.float(), b.float(), 1, 1) + tv.float()[:, :, None, None] + res.float()
    xg, wg = x.permute(0, 2, 3, 1).contiguous().to(dev), ops.conv_weight_okc(w).to(dev)
    got = ops.conv2d_nhwc_f16(xg, wg, b.to(dev), 3, 1, 1, False, tv.to(dev),
                              res.permute(0, 2, 3, 1).contiguous().to(dev), split_k=split)
    torch.testing.assert_close(got.cpu().float().permute(0, 3, 1, 2), ref, rtol=3e-3, atol=3e-3)
    w6 = _r((6, C, 3, 3), 19, (C * 9) ** -0.5)
    ref6 = F.conv2d(x.float(), w6.float(), None, 1, 1)
    got6 = ops.conv2d_nhwc_f16(xg, ops.conv_weight_okc(w6).to(dev), None, 3, 1, 1, split_k=split)
    torch.testing.assert_close(got6.cpu().float().permute(0, 3, 1, 2), ref6, rtol=3e-3, atol=3e-3)


def test_conv_f16_split_k_heuristic():
    """Host-only: the library splits K only where the output tiles leave the chip idle."""
    from drawingspinup_amd._lib import lib
    f = lib().dsu_conv2d_nhwc_f16_split_k
    assert f(12, 32, 32, 320, 320, 3, 1, 1, 0) == 1          # 96 x 3 tiles: plain kernel
    assert f(12, 4, 4, 1280, 1280, 3, 1, 1, 0) >= 8          # 2 x 10 tiles, 180 chunks
    assert f(12, 8, 8, 1280, 1280, 3, 1, 1, 0) >= 4
    assert f(12, 4, 4, 64, 1280, 1, 1, 0, 0) == 1            # 1 chunk: nothing to split
