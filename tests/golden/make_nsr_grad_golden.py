"""Generate tests/golden/nsr_grad_reference.npz: the parameter gradients of ONE optimisation step
as the REFERENCE computes them — OrthoNeuSSystem.training_step (instant_nsr/systems/neus_ortho.py:
79-169) over NeuSModelTextureMLP.forward (models/neus.py:114-196), then `loss.backward()`.

    python tests/golden/make_nsr_grad_golden.py        # needs /root/reference (~2 min)

Everything between the loss and the parameters is the reference's own code and torch autograd
(VanillaMLP with weight norm, VarianceNetwork, the finite-difference VolumeSDF.forward, get_alpha,
the composites, F.normalize, the loss section).  The two third-party native ops are served by the
oracle restatements made differentiable for this script:

  * tinycudann.Encoding: forward = oracle/hashgrid.encode, backward w.r.t. the table =
    oracle/hashgrid.encode_bwd (float64 scatter-add, no loss scaling);
  * nerfacc.render_weight_from_alpha: forward/backward = oracle/nerfacc_ref (the published
    backward with max(1 - alpha, 1e-10)); accumulate_along_rays = index_add (autograd's own
    backward), ray_marching as in make_nsr_step_golden.py.

Inputs are those of make_nsr_step_golden.py (same model, rays, batch, injected draws), so the
forward values of this run equal nsr_step_reference.npz (checked below).  Stored: the gradient of
every parameter (the 7.7 M-float hash table as (index, value) of its non-zeros).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_nsr_step_golden as S  # noqa: E402
from oracle import hashgrid as oh, nerfacc_ref as nr  # noqa: E402


class _EncFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, enc):
        tab = params.detach().half().numpy().reshape(-1, 2)
        ctx.enc, ctx.x = enc, x.detach().float().numpy()
        return torch.from_numpy(oh.encode(tab, ctx.x, enc.lv, enc.n_levels))

    @staticmethod
    def backward(ctx, dout):
        d = dout.detach().double().numpy()
        live = [l for l in range(ctx.enc.n_levels) if np.any(d[:, 2 * l:2 * l + 2])]
        active = (max(live) + 1) if live else 0
        g = oh.encode_bwd(ctx.x, d, ctx.enc.lv, active)
        return None, torch.from_numpy(g.reshape(-1)).float(), None


class EncDiff(S._Enc):
    def forward(self, x):
        return _EncFn.apply(x, self.params, self)


class _WeightFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha, counts):
        a = alpha.detach().numpy().reshape(-1)
        ctx.a, ctx.c = a, counts
        return torch.from_numpy(nr.render_weight_from_alpha(a, counts).astype(np.float32))[:, None]

    @staticmethod
    def backward(ctx, gw):
        ga = nr.render_weight_from_alpha_bwd(ctx.a, ctx.c, gw.detach().numpy().reshape(-1))
        return torch.from_numpy(ga.astype(np.float32)).view(-1, 1), None


def render_weight_from_alpha(alpha, ray_indices=None, n_rays=None):
    return _WeightFn.apply(alpha, S._counts(ray_indices, n_rays))


def accumulate_along_rays(weights, ray_indices, values=None, n_rays=None):
    src = weights if values is None else weights * values
    out = torch.zeros(n_rays, src.shape[-1], dtype=src.dtype)
    return out.index_add(0, ray_indices.reshape(-1).long(), src)


if __name__ == "__main__":
    S.install_stubs()
    sys.modules["tinycudann"].Encoding = EncDiff
    sys.modules["nerfacc"].render_weight_from_alpha = render_weight_from_alpha
    sys.modules["nerfacc"].accumulate_along_rays = accumulate_along_rays
    sys.path.insert(0, S.REF)
    from drawingspinup_amd.nsr.model import Cfg, DEFAULT_MODEL_CONFIG
    from drawingspinup_amd.nsr.system import DEFAULT_SYSTEM_CONFIG
    import instant_nsr.systems.utils  # noqa: F401
    from instant_nsr import models as ref_models
    from instant_nsr.systems.neus_ortho import OrthoNeuSSystem as System

    STEP_GOLD = np.load(os.path.join(HERE, "nsr_step_reference.npz"))
    torch.manual_seed(0)
    cfg = Cfg(DEFAULT_MODEL_CONFIG)
    cfg["randomized"] = False
    model = ref_models.make("neus", cfg)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in model.geometry.network.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.05)
        for p in model.texture.network.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.05)
    # same model as the step fixture (make_nsr_step_golden.py builds it the same way)
    for k, v in model.state_dict().items():
        if "sd." + k in STEP_GOLD.files:
            assert np.array_equal(v.numpy(), STEP_GOLD["sd." + k]), k
    model.occupancy_grid.binary = torch.from_numpy(S.shell_occupancy())
    STEP = int(STEP_GOLD["step"])
    model.train()
    model.randomized = False
    rays = torch.from_numpy(STEP_GOLD["rays"])
    batch = {k[6:]: torch.from_numpy(STEP_GOLD[k]) for k in STEP_GOLD.files if k.startswith("batch.")}

    logged = {}
    sysm = object.__new__(System)
    torch.nn.Module.__init__(sysm)
    lossc = Cfg(dict(DEFAULT_SYSTEM_CONFIG.loss))
    sysm.config = Cfg({"model": {"dynamic_ray_sampling": True, "max_train_num_rays": 8192},
                       "system": {"loss": dict(lossc)}})
    sysm.train_num_rays, sysm.train_num_samples = 256, 256 * 1024
    sysm.global_step_, sysm.current_epoch_ = STEP, 0
    System.global_step = property(lambda self: self.global_step_)
    System.current_epoch = property(lambda self: self.current_epoch_)
    sysm.log = lambda name, value, **k: logged.__setitem__(name, value)

    class _DS:
        has_mask = True
    sysm.dataset = _DS()
    sysm.model = model
    model.update_step(0, STEP)
    kept = {}

    def fwd(b):
        torch.manual_seed(5)                                   # DRAW_SEED of the step fixture
        out = model(b["rays"])
        kept.update(out)
        return out
    sysm.forward = fwd
    _sort = torch.sort
    torch.sort = lambda x, *a, **k: _sort(x, *a, **{**k, "stable": True})     # see make_nsr_step_golden.py
    res = sysm.training_step({k: v.clone() for k, v in batch.items()}, 0)
    torch.sort = _sort
    for k in ("sdf_samples", "random_sdf", "weights"):
        assert np.array_equal(kept[k].detach().numpy(), STEP_GOLD["fwd." + k]), k
    for k in ("comp_rgb", "opacity", "comp_normal"):           # f32 index_add here, f64 add.at there
        np.testing.assert_allclose(kept[k].detach().numpy(), STEP_GOLD["fwd." + k], rtol=0, atol=2e-6)
    assert abs(float(res["loss"]) - float(STEP_GOLD["loss.total"])) < 1e-6 * abs(float(STEP_GOLD["loss.total"]))
    res["loss"].backward()
    out = {"loss.total": res["loss"].detach().numpy()}
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        gnp = p.grad.numpy()
        if p.numel() > 100000:
            nz = np.flatnonzero(gnp)
            out["gradnz_idx." + name] = nz.astype(np.int32)
            out["gradnz_val." + name] = gnp.reshape(-1)[nz]
            out["gradnz_numel." + name] = np.int64(p.numel())
        else:
            out["grad." + name] = gnp
    np.savez_compressed(os.path.join(HERE, "nsr_grad_reference.npz"), **out)
    print("wrote nsr_grad_reference.npz")
    for k, v in out.items():
        print(k, v.shape, float(np.abs(v).max()) if v.size else 0.0)
