"""The drop-in import names resolve to the gfx950 kernels and keep the third-party semantics."""
import importlib
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def shims():
    from drawingspinup_amd import shims as s
    path = s.install()
    saved = {k: sys.modules.pop(k) for k in list(sys.modules)
             if k.split(".")[0] in ("tinycudann", "nerfacc", "xformers", "torchvision")}
    yield path
    for k in list(sys.modules):
        if k.split(".")[0] in ("tinycudann", "nerfacc", "xformers", "torchvision"):
            sys.modules.pop(k)
    sys.modules.update(saved)
    sys.path.remove(path)


def test_tinycudann_encoding_module(dev, shims):
    import tinycudann as tcnn
    from oracle import hashgrid as oh
    cfg = {"otype": "HashGrid", "n_levels": 10, "n_features_per_level": 2, "log2_hashmap_size": 19,
           "base_resolution": 32, "per_level_scale": 1.3195079107728942}
    enc = tcnn.Encoding(3, cfg).to(dev)
    assert enc.n_output_dims == 20 and enc.params.shape == (3838848 * 2,)
    assert float(enc.params.abs().max()) <= 1e-4                    # tcnn init range
    with torch.no_grad():
        enc.params.mul_(2000.0)
    x = torch.rand(1000, 3, device=dev)
    y = enc(x)
    assert y.dtype == torch.float16 and y.shape == (1000, 20)
    ref = oh.encode(enc.params.detach().half().cpu().numpy().reshape(-1, 2), x.cpu().numpy(),
                    oh.make_levels(), 10)
    assert np.array_equal(y.detach().cpu().numpy().view(np.uint16), ref.view(np.uint16))
    (y.float() * torch.randn_like(y.float())).sum().backward()
    assert enc.params.grad is not None and enc.params.grad.shape == enc.params.shape
    assert int((enc.params.grad != 0).sum()) > 0
    tcnn.free_temporary_memory()


def test_xformers_and_nerfacc_and_torchvision_names(dev, shims):
    import xformers.ops
    from oracle import mv_ref as mr
    g = torch.Generator().manual_seed(0)
    q, k, v = [torch.randn(16, 64, 40, generator=g).half() for _ in range(3)]
    out = xformers.ops.memory_efficient_attention(q.to(dev), k.to(dev), v.to(dev))
    ref = mr.memory_efficient_attention(q.double(), k.double(), v.double())
    torch.testing.assert_close(out.cpu().double(), ref, rtol=2e-3, atol=2e-3)
    nerfacc = importlib.import_module("nerfacc")
    for name in ("ContractionType", "OccupancyGrid", "ray_marching", "render_weight_from_alpha",
                 "accumulate_along_rays"):
        assert hasattr(nerfacc, name)
    grid = nerfacc.OccupancyGrid(roi_aabb=[-1, -1, -1, 1, 1, 1], resolution=128,
                                 contraction_type=nerfacc.ContractionType.AABB).to(dev)
    grid.train()
    grid.every_n_step(step=0, occ_eval_fn=lambda x: (x.norm(dim=-1, keepdim=True) < 0.5).float(),
                      occ_thre=0.001)
    frac = float(grid.binary.float().mean())
    assert abs(frac - (4 / 3 * np.pi * 0.125) / 8) < 0.01           # sphere r=0.5 in [-1,1]^3
    rays_o = torch.tensor([[0.0, 0.0, -1.3]], device=dev)
    rays_d = torch.tensor([[0.0, 0.0, 1.0]], device=dev)
    ri, ts, te = nerfacc.ray_marching(rays_o, rays_d, scene_aabb=torch.tensor([-1., -1, -1, 1, 1, 1]),
                                      grid=grid, render_step_size=1.732 * 2 / 1024, stratified=False)
    assert ts.shape[1] == 1 and abs(float((te[-1] - ts[0])) - 1.0) < 0.02   # chord through the sphere
    w = nerfacc.render_weight_from_alpha(torch.full((ri.numel(), 1), 0.5, device=dev),
                                         ray_indices=ri, n_rays=1)
    op = nerfacc.accumulate_along_rays(w, ri, values=None, n_rays=1)
    assert abs(float(op) - 1.0) < 1e-4
    from torchvision.ops import deform_conv2d
    x, wt = torch.randn(1, 8, 16, 16, device=dev), torch.randn(4, 8, 3, 3, device=dev)
    y = deform_conv2d(input=x, offset=torch.zeros(1, 18, 16, 16, device=dev), weight=wt, padding=(1, 1))
    torch.testing.assert_close(y, torch.nn.functional.conv2d(x, wt, padding=1), rtol=1e-4, atol=1e-4)
