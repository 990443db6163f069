"""gfx950 UNet (NHWC f16, HIP conv/attention/norm kernels) vs the float64 functional oracle."""
import pytest
import torch

from drawingspinup_amd import ops
from drawingspinup_amd.mv.unet import UNetMV2DConditionModel
from oracle import mv_ref as mr

pytestmark = pytest.mark.gpu

SMALL = dict(block_out_channels=(320, 640, 640), layers_per_block=1,
             down_block_types=("CrossAttnDownBlockMV2D", "CrossAttnDownBlockMV2D", "DownBlock2D"),
             up_block_types=("UpBlock2D", "CrossAttnUpBlockMV2D", "CrossAttnUpBlockMV2D"))


def _init(model, seed):
    g = torch.Generator().manual_seed(seed)
    for n, p in model.named_parameters():
        if p.dim() > 1:
            fan_in = p[0].numel()
            p.data = torch.randn(p.shape, generator=g) * fan_in ** -0.5
        elif "norm" in n and n.endswith("weight"):
            p.data = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        else:
            p.data = 0.05 * torch.randn(p.shape, generator=g)
    # the zero-initialised joint to_out (transformer_mv2d.py:499,516) would hide that branch
    return model


def test_norm_kernels(dev):
    g = torch.Generator().manual_seed(0)
    for C in (320, 640, 1920, 2560):
        x = torch.randn(3, 5, 7, C, generator=g).half()
        w, b = (1 + 0.1 * torch.randn(C, generator=g)).half(), (0.1 * torch.randn(C, generator=g)).half()
        for silu in (False, True):
            ref = torch.nn.functional.group_norm(x.float().permute(0, 3, 1, 2), 32, w.float(),
                                                 b.float(), 1e-5)
            ref = torch.nn.functional.silu(ref) if silu else ref
            got = ops.groupnorm_nhwc_f16(x.to(dev), w.to(dev), b.to(dev), 32, 1e-5, silu)
            torch.testing.assert_close(got.cpu().float().permute(0, 3, 1, 2), ref, rtol=2e-3, atol=2e-3)
    for C in (320, 640, 1280):
        x = torch.randn(37, C, generator=g).half()
        w, b = (1 + 0.1 * torch.randn(C, generator=g)).half(), (0.1 * torch.randn(C, generator=g)).half()
        ref = torch.nn.functional.layer_norm(x.float(), (C,), w.float(), b.float(), 1e-5)
        torch.testing.assert_close(ops.layernorm_f16(x.to(dev), w.to(dev), b.to(dev)).cpu().float(),
                                   ref, rtol=2e-3, atol=2e-3)
    h = torch.randn(19, 2 * 1280, generator=g).half()
    a, gt = h.float().chunk(2, -1)
    torch.testing.assert_close(ops.geglu_f16(h.to(dev)).cpu().float(),
                               a * torch.nn.functional.gelu(gt), rtol=2e-3, atol=2e-3)


def test_unet_small_vs_oracle(dev):
    torch.manual_seed(0)
    model = _init(UNetMV2DConditionModel(**SMALL), 1).half()
    sd = {k: v.clone() for k, v in model.state_dict().items()}     # f16 values, shared by both
    g = torch.Generator().manual_seed(2)
    sample = torch.randn(12, 8, 8, 8, generator=g).half()
    ctx = torch.randn(12, 1, 768, generator=g).half()
    cl = torch.randn(12, 10, generator=g).half()
    t = torch.tensor([487])
    ref = mr.UNetRef(sd, SMALL["block_out_channels"], SMALL["down_block_types"],
                     SMALL["up_block_types"], layers_per_block=1)(sample, t, ctx, cl)
    out = model.to(dev)(sample.to(dev), t.to(dev), ctx.to(dev), cl.to(dev)).cpu().double()
    rel = (out - ref).norm() / ref.norm()
    print("unet rel-L2", float(rel))
    assert out.shape == (12, 4, 8, 8)
    assert float(rel) < 5e-3           # f16 activations through ~40 layers vs float64
