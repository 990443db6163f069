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
@pytest.mark.parametrize("fixup", [True, False])
@pytest.mark.parametrize("split", [1, 2, 7, 16])
def test_conv_f16_split_k_matches_plain(dev, split, fixup, monkeypatch):
    """Split-K (f32 partials, summed by the last workgroup of a tile — `fixup` — or by the separate
    reduce/epilogue kernel) against the single-pass kernel and torch, on a 4x4-level UNet shape with
    the full fused epilogue and on an O that is not a multiple of 4.  The fix-up form is run
    three times on the same counters: they must come back to zero."""
    monkeypatch.setattr(ops, "SPLITK_FIXUP", fixup)
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
    if fixup and split > 1:
        for _ in range(2):
            again = ops.conv2d_nhwc_f16(xg, wg, b.to(dev), 3, 1, 1, False, tv.to(dev),
                                        res.permute(0, 2, 3, 1).contiguous().to(dev), split_k=split)
            assert torch.equal(again, got)                    # same z-order sum whoever arrives last
        assert int(ops._tile_counters(dev).abs().sum()) == 0


def test_conv_f16_split_k_heuristic():
    """Host-only: the library splits K only where the output tiles leave the chip idle."""
    from drawingspinup_amd._lib import lib
    f = lib().dsu_conv2d_nhwc_f16_split_k
    # (64 x 64 tiles for these shapes; the split aims at half of the 512 workgroups that fill the chip)
    assert f(12, 32, 32, 320, 320, 3, 1, 1, 0) == 1          # 192 x 5 tiles: plain kernel
    s44 = f(12, 4, 4, 1280, 1280, 3, 1, 1, 0)                # 3 x 20 tiles, 180 chunks
    assert 4 <= s44 <= 16 and 60 * s44 >= 256
    s88 = f(12, 8, 8, 1280, 1280, 3, 1, 1, 0)                # 12 x 20 tiles
    assert 2 <= s88 <= 4 and 240 * s88 >= 256
    assert f(12, 4, 4, 64, 1280, 1, 1, 0, 0) == 1            # 1 chunk: nothing to split
    assert f(12, 64, 64, 320, 320, 3, 1, 1, 0) == 1          # 128 x 128 tiles (1152 of them), no split


# ---------------------------------------------------------------------------------------------
# The UNet's linear layers on the same MFMA kernel (dsu_gemm_f16_fwd / dsu_gemm_geglu_fwd)
# ---------------------------------------------------------------------------------------------
@gpu
@pytest.mark.parametrize("M,K,N", [(12 * 1024, 320, 320), (12 * 256, 640, 640), (12 * 16, 1280, 1280),
                                   (12, 768, 1280), (12, 1280, 320), (37, 16, 1280), (12 * 64, 1280, 1280)])
def test_linear_f16_matches_torch(dev, M, K, N):
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).half()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half()
    b = (torch.randn(N, generator=g) * 0.1).half()
    r = torch.randn(M, N, generator=g).half()
    ref = x.float() @ w.float().t() + b.float() + r.float()
    got = ops.linear_f16(x.to(dev), w.to(dev), b.to(dev), residual=r.to(dev)).cpu().float()
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=4e-3)
    got = ops.linear_f16(x.to(dev), w.to(dev)).cpu().float()             # no bias, no residual
    torch.testing.assert_close(got, x.float() @ w.float().t(), rtol=2e-3, atol=4e-3)
    for sk in (2, 5):                                                       # explicit split-K
        if K >= 64 * sk:
            got = ops.linear_f16(x.to(dev), w.to(dev), b.to(dev), residual=r.to(dev), split_k=sk)
            torch.testing.assert_close(got.cpu().float(), ref, rtol=2e-3, atol=4e-3)


@gpu
@pytest.mark.parametrize("B,T,K,N", [(12, 1024, 320, 320), (12, 16, 1280, 1280), (3, 100, 640, 648)])
def test_linear_f16_transposed_output_is_v_transposed(dev, B, T, K, N):
    g = torch.Generator().manual_seed(B + T + K)
    x = torch.randn(B, T, K, generator=g).half()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half()
    got = ops.linear_f16(x.to(dev), w.to(dev), transposed_tokens=T).cpu().float()
    ref = torch.matmul(w.float(), x.float().transpose(1, 2))               # (B, N, T)
    assert got.shape == (B, N, T)
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=4e-3)


@gpu
@pytest.mark.parametrize("M,C", [(12 * 1024, 320), (12 * 64, 1280), (19, 640), (130, 328)])
def test_linear_geglu_matches_unfused_reference(dev, M, C):
    """diffusers GEGLU: proj -> chunk(2) -> a * gelu(g), the projection rounded to f16 first."""
    g = torch.Generator().manual_seed(M + C)
    N = 4 * C
    x = torch.randn(M, C, generator=g).half()
    w = (torch.randn(2 * N, C, generator=g) * C ** -0.5).half()
    b = (torch.randn(2 * N, generator=g) * 0.1).half()
    proj = (x.float() @ w.float().t() + b.float()).half().float()
    a, gate = proj.chunk(2, -1)
    ref = a * torch.nn.functional.gelu(gate)
    got = ops.linear_geglu_f16(x.to(dev), w.to(dev), b.to(dev)).cpu().float()
    assert got.shape == (M, N)
    # one f16 ulp of the projection can flip under a different accumulation order
    torch.testing.assert_close(got, ref, rtol=4e-3, atol=6e-3)
    fused_old = ops.geglu_f16((x.to(dev) @ w.to(dev).t() + b.to(dev)).contiguous()).cpu().float()
    torch.testing.assert_close(got, fused_old, rtol=4e-3, atol=6e-3)
