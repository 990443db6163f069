"""Fused per-step kernels of the NSR optimisation loop (csrc/nsr_step.hip) against the op-by-op
torch formulation that tests/test_host_logic.py pins to the reference's criterions, and the
hand-sequenced training step against the autograd step."""
import numpy as np
import pytest
import torch

from drawingspinup_amd import ops
from drawingspinup_amd.nsr.system import (DEFAULT_SYSTEM_CONFIG, Cfg, OrthoData, OrthoNeuSSystem)

pytestmark = pytest.mark.gpu


class _LossOnly(OrthoNeuSSystem):
    """OrthoNeuSSystem.ray_losses / sample_losses without building a model."""

    def __init__(self, loss_cfg, has_mask=True):
        self.config = Cfg({"loss": loss_cfg})

        class _D:
            pass
        self.dataset = _D()
        self.dataset.has_mask = has_mask


def _ray_batch(R, seed, dev):
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g)
    comp = torch.zeros(R, 8)
    comp[:, 0] = u(R) * 1.2 - 0.1                       # opacity incl. values beyond both clamps
    comp[: R // 5, 0] = 1e-4                            # tied smallest BCE errors (mask 0 below)
    comp[:, 1] = u(R) * 2
    comp[:, 2:5] = u(R, 3)
    comp[:, 5:8] = (u(R, 3) * 2 - 1) * u(R, 1)
    mask = (u(R) > 0.4).float()
    mask[: R // 5] = 0.0
    comp[R - 7:, 5:8] = 0.0                             # rays without samples: zero composite ...
    mask[R - 7:] = 0.0                                  # ... are never foreground here
    nrm = torch.nn.functional.normalize(u(R, 3) * 2 - 1, dim=-1)
    batch = {"rgb": u(R, 3), "normal": nrm, "mask": mask, "cosines": u(R) * 1.2 - 1.0,
             "view_weights": u(R) + 0.5}
    return comp.to(dev), {k: v.to(dev) for k, v in batch.items()}


@pytest.mark.parametrize("R,geo,l1", [(1500, True, 0.0), (1024, False, 0.3), (2049, True, 0.2),
                                      (37, True, 0.0), (8192, True, 0.0)])
def test_ray_losses_kernel_matches_torch_path(dev, R, geo, l1):
    L = dict(DEFAULT_SYSTEM_CONFIG.loss)
    L.update(geo_aware=geo, lambda_rgb_l1=l1)
    ref = _LossOnly(L)
    comp, batch = _ray_batch(R, 11 + R, dev)
    c = comp.clone().requires_grad_(True)
    terms = ref.ray_losses(c, batch)
    (g_ref,) = torch.autograd.grad(sum(terms.values()), c)
    t, d = ops.ray_losses(comp, batch["rgb"], batch["normal"], batch["mask"], batch["cosines"],
                          batch["view_weights"],
                          {"rgb_p_ratio": L["rgb_p_ratio"], "normal_p_ratio": L["normal_p_ratio"],
                           "mask_p_ratio": L["mask_p_ratio"], "lambda_rgb_mse": L["lambda_rgb_mse"],
                           "lambda_rgb_l1": l1, "lambda_normal": L["lambda_normal"],
                           "lambda_mask": L["lambda_mask"], "geo_aware": geo})
    t = t.cpu()
    # float tolerance: same selection, different summation order
    np.testing.assert_allclose(float(t[0]), float(terms["rgb_mse"]), rtol=2e-5, atol=1e-7)
    if l1:
        np.testing.assert_allclose(float(t[1]), float(terms["rgb_l1"]), rtol=2e-5, atol=1e-7)
    else:
        assert float(t[1]) == 0.0
    np.testing.assert_allclose(float(t[2]), float(terms["normal"]), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(float(t[3]), float(terms["mask"]), rtol=2e-5, atol=1e-7)
    assert float(d[:, 1].abs().max()) == 0.0
    scale = float(g_ref.abs().max())
    assert float((d - g_ref).abs().max()) < 2e-5 * scale + 1e-9


def test_ray_losses_kernel_empty_foreground_is_nan_like_the_reference(dev):
    """k = int(ratio * 0) = 0: torch.mean over an empty selection is NaN in the reference; the
    gradient is zero."""
    L = dict(DEFAULT_SYSTEM_CONFIG.loss)
    comp, batch = _ray_batch(256, 5, dev)
    batch["mask"].zero_()
    t, d = ops.ray_losses(comp, batch["rgb"], batch["normal"], batch["mask"], batch["cosines"],
                          batch["view_weights"], {**L, "lambda_rgb_l1": 0.0})
    assert torch.isnan(t[0]) and torch.isfinite(t[3])
    assert float(d[:, 2:8].abs().max()) == 0.0


@pytest.mark.parametrize("n_s,n_r,smooth", [(100000, 2048, 1.0), (777, 64, 0.0), (0, 2048, 1.0)])
def test_sample_losses_kernel_matches_torch_path(dev, n_s, n_r, smooth):
    L = dict(DEFAULT_SYSTEM_CONFIG.loss)
    L["lambda_3d_normal_smooth"] = smooth
    ref = _LossOnly(L)
    g = torch.Generator().manual_seed(3 + n_s)
    n = n_s + 2 * n_r
    sdf = (torch.randn(n, generator=g) * 0.02).to(dev)
    grad = (torch.randn(n, 3, generator=g) * 0.7).to(dev)
    if n_s:
        grad[0] = 0.0                                    # |grad| = 0: subgradient 0
    sdf_r, grad_r = sdf.clone().requires_grad_(True), grad.clone().requires_grad_(True)
    out = {"sdf_grad_samples": grad_r[:n_s], "random_sdf": sdf_r[n_s:n_s + n_r],
           "random_sdf_grad": grad_r[n_s:n_s + n_r], "normal_perturb": grad_r[n_s + n_r:]}
    if n_s == 0:
        out["sdf_grad_samples"] = grad_r[:1] * 0 + 1     # empty mean is NaN in torch; kernel: 0
    terms = ref.sample_losses(out)
    gs, gg = torch.autograd.grad(sum(terms.values()), [sdf_r, grad_r], allow_unused=True)
    gs = torch.zeros_like(sdf) if gs is None else gs
    t, d_sdf, d_grad = ops.sample_losses(sdf, grad, n_s, n_r, L["lambda_eikonal"],
                                         L["lambda_sparsity"], L["sparsity_scale"], smooth)
    if n_s:
        np.testing.assert_allclose(float(t[0]), float(terms["eikonal"]), rtol=1e-4)
    np.testing.assert_allclose(float(t[1]), float(terms["sparsity"]), rtol=1e-4)
    if smooth:
        np.testing.assert_allclose(float(t[2]), float(terms["normal_smooth"]), rtol=1e-4)
    torch.testing.assert_close(d_sdf, gs, rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(d_grad, gg, rtol=1e-5, atol=1e-9)
    # accumulate form: prefix rows keep what is there and receive the eikonal part on top
    pre_s = torch.full((n,), 3.0, device=dev)
    pre_g = torch.full((n, 3), -2.0, device=dev)
    ops.sample_losses(sdf, grad, n_s, n_r, L["lambda_eikonal"], L["lambda_sparsity"],
                      L["sparsity_scale"], smooth, pre_s, pre_g)
    torch.testing.assert_close(pre_s[:n_s], torch.full((n_s,), 3.0, device=dev))
    torch.testing.assert_close(pre_g[:n_s], d_grad[:n_s] - 2.0, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(pre_s[n_s:], d_sdf[n_s:])
    torch.testing.assert_close(pre_g[n_s:], d_grad[n_s:])


def test_march_points_matches_packed_march(dev):
    """ray_march_points = ray_march_single_pass + the reference's position arithmetic, bit for
    bit, with the offsets scan kernel instead of cumsum."""
    g = torch.Generator().manual_seed(2)
    n = 3000
    o = (torch.rand(n, 3, generator=g) * 2 - 1) * 0.3
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    o = (o - 2.0 * d).to(dev)
    d = d.to(dev)
    res = 64
    occ = (torch.rand(res ** 3, generator=g) < 0.3).to(torch.uint8).to(dev)
    aabb = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]
    step = 1.732 * 2 / 512
    tmin, tmax = ops.ray_aabb(o, d, aabb, None, step)
    ri, ts, te, off, cnt = ops.ray_march_single_pass(o, d, tmin, tmax, aabb, occ, res, step)
    pts, ts2, te2, off2, cnt2, total = ops.ray_march_points(o, d, tmin, tmax, aabb, occ, res, step,
                                                            tail_rows=5)
    assert total == ri.shape[0] > 1000 and pts.shape == (total + 5, 3)
    assert torch.equal(off, off2) and torch.equal(cnt, cnt2)
    assert torch.equal(ts, ts2) and torch.equal(te, te2)
    mid = (ts + te) / 2.0
    ref = o[ri] + d[ri] * mid[:, None]
    assert torch.equal(pts[:total], ref)


def test_fused_step_gradients_match_autograd_step(dev):
    """Same parameters, same injected draws: every parameter gradient of the hand-sequenced step
    equals the autograd step's (learning rate 0 so that the step leaves the parameters alone)."""
    from test_gpu_nsr_model import _inject, _loss_and_grads
    ds = OrthoData.synthetic_sphere(256, device=dev)
    sysm = OrthoNeuSSystem(device=dev, seed=7)
    sysm.dataset = ds
    for s in range(20):
        sysm.train_num_rays = 256
        sysm.training_step(_inject(sysm, ds, 256, 100 + s, dev))
    sysm.global_step = 17
    sysm.train_num_rays = 512
    sysm._base_lrs = [0.0 for _ in sysm._base_lrs]
    inj = _inject(sysm, ds, 512, 999, dev)
    loss_a, out_a, g_a = _loss_and_grads(sysm, True, inj)
    sysm.train_num_rays = 512
    sysm.keep_table_grad = True
    last = sysm.training_step_fused(inj)
    sysm.global_step = 17
    g_f = {n: p.grad.detach().clone() for n, p in sysm.model.named_parameters()
           if p.grad is not None}
    assert last["n_samples"] == int(out_a["num_samples"])
    assert abs(float(last["loss"]) - loss_a) < 2e-5 * max(1.0, abs(loss_a))
    assert set(g_f) == set(g_a)
    for n in g_a:
        scale = float(g_a[n].abs().max()) + 1e-12
        err = float((g_f[n] - g_a[n]).abs().max()) / scale
        assert err < 1e-3, (n, err)


@pytest.mark.parametrize("n", [1, 63, 64, 1000, 262144 + 37])
def test_texture_mlp_kernels_match_torch(dev, n):
    """sigmoid(VanillaMLP 16->64->64->3) forward and backward (f32 MFMA) vs torch f32
    (tolerance: different f32 summation order only; reference texture.py:20-30)."""
    g = torch.Generator().manual_seed(n)
    mk = lambda *s: (torch.randn(*s, generator=g)).to(dev)
    params = [mk(64, 16) * 0.3, mk(64) * 0.1, mk(64, 64) * 0.15, mk(64) * 0.1, mk(3, 64) * 0.2,
              mk(3) * 0.1]
    x = mk(n, 16)
    d_rgb = mk(n, 3)
    px = [p.clone().requires_grad_(True) for p in params]
    xr = x.clone().requires_grad_(True)
    h = torch.relu(torch.nn.functional.linear(xr, px[0], px[1]))
    h = torch.relu(torch.nn.functional.linear(h, px[2], px[3]))
    ref = torch.sigmoid(torch.nn.functional.linear(h, px[4], px[5]))
    ref.backward(d_rgb)
    rgb = ops.texture_fwd(params, x)
    torch.testing.assert_close(rgb, ref.detach(), rtol=1e-5, atol=2e-6)
    d_x, gp = ops.texture_bwd(params, x, rgb, d_rgb)
    torch.testing.assert_close(d_x, xr.grad, rtol=1e-4, atol=1e-5 * float(xr.grad.abs().max()))
    for got, p in zip(gp, px):
        scale = float(p.grad.abs().max()) + 1e-12
        assert float((got - p.grad).abs().max()) < 2e-4 * scale, (tuple(p.shape), scale)


@pytest.mark.parametrize("n", [1, 97, 4096 + 33, 262144 + 37])
def test_texture_backward_with_the_forwards_relu_pattern(dev, n):
    """dsu_texture_fwd_shaded_m hands the ReLU pattern of hidden layer 1 to the backward
    (dsu_texture_bwd_shaded_partials_m), whose recompute of that layer then runs as bf16 x 3: the
    pattern is the exact forward's (bit for bit against a torch forward away from |pre| < 1e-6), and
    every gradient stays within the bf16 x 3 bound (2e-4 of its largest entry here; measured ~1e-5
    rel-L2) of the exact-recompute path — what the NSR step driver runs."""
    g = torch.Generator().manual_seed(1000 + n)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)
    params = [mk(64, 16) * 0.3, mk(64) * 0.1, mk(64, 64) * 0.15, mk(64) * 0.1, mk(3, 64) * 0.2, mk(3) * 0.1]
    feat, grad = mk(n, 13) * 0.5, mk(n, 3)
    d_rgb, d_nrm = mk(n, 3), mk(n, 3) * 0.1
    normal, rgb = ops.texture_fwd_shaded(params, feat, grad)
    normal2, rgb2, mask = ops.texture_fwd_shaded(params, feat, grad, with_mask=True)
    assert torch.equal(rgb, rgb2) and torch.equal(normal, normal2)
    # the pattern against a torch forward
    x = torch.cat([feat, torch.nn.functional.normalize(grad, dim=-1)], -1)
    h0 = torch.relu(torch.nn.functional.linear(x, params[0], params[1]))
    pre1 = torch.nn.functional.linear(h0, params[2], params[3])
    unit = torch.arange(64, device=dev)
    hh, T, low = (unit >> 2) & 1, unit >> 5, unit & 31
    r = (low & 3) + 4 * (low >> 3)                       # unit = 32 T + (r & 3) + 8 (r >> 2) + 4 h
    bits = (mask[:, hh].long() >> (16 * T + r)) & 1      # (n, 64)
    clear = pre1.abs() > 1e-5
    assert torch.equal(bits[clear] == 1, (pre1 > 0)[clear])
    a = ops.texture_bwd_shaded_partials(params, feat, grad, rgb, d_rgb, d_nrm, 8)
    b = ops.texture_bwd_shaded_partials(params, feat, grad, rgb, d_rgb, d_nrm, 8, h1_mask=mask)
    for k in range(2):
        tol = 2e-4 * float(a[k].abs().max()) + 1e-12
        assert float((a[k] - b[k]).abs().max()) <= tol, k
    assert not bool(b[1][n:].any())                      # tail rows zeroed
    # the deferred parameter sums (carried out here by the reduction the scatter launch would do)
    ga = ops.texture_bwd_shaded(params, feat, grad, rgb, d_rgb, d_nrm, 0)[2]
    from drawingspinup_amd.ops import texture_partial_map
    tmap = torch.from_numpy(texture_partial_map()).to(dev).long()
    rec = b[3].record
    nb, stride, npart = int(rec.nblocks), int(rec.stride), int(rec.n)
    ws = b[3].keep[0][:nb * stride].view(nb, stride)[:, :npart].sum(0)
    flat = torch.zeros(sum(t.numel() for t in params), device=dev)
    flat.index_add_(0, tmap[tmap >= 0], ws[tmap >= 0])
    off = 0
    for t, gx in zip(params, ga):
        got = flat[off:off + t.numel()].view_as(t)
        off += t.numel()
        assert float((got - gx).abs().max()) <= 2e-4 * float(gx.abs().max()) + 1e-12, tuple(t.shape)


def test_texture_autograd_function_in_model(dev):
    from drawingspinup_amd.nsr.model import DEFAULT_MODEL_CONFIG, VolumeRadiance
    tex = VolumeRadiance(DEFAULT_MODEL_CONFIG.texture).to(dev)
    assert tex.fused_ok
    x = torch.randn(5000, 16, device=dev)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y1 = tex.rgb_fused(x1)
    y2 = torch.sigmoid(tex.network(x2))
    torch.testing.assert_close(y1, y2, rtol=1e-5, atol=2e-6)
    gy = torch.randn_like(y1)
    g1 = torch.autograd.grad(y1, [x1] + list(tex.parameters()), gy)
    g2 = torch.autograd.grad(y2, [x2] + list(tex.parameters()), gy)
    for a, b in zip(g1, g2):
        assert float((a - b).abs().max()) < 2e-4 * (float(b.abs().max()) + 1e-12)


def test_fused_ray_batch_matches_op_by_op_preprocess(dev):
    """dsu_ortho_ray_batch against the op-by-op preprocess_data (neus_ortho.py:26-82) on the same
    (view, y, x) draws: gathers exact, ray arithmetic to f32 rounding."""
    ds = OrthoData.synthetic_sphere(128, device=dev)
    ds.view_weights = torch.rand_like(ds.view_weights) + 0.5
    sysm = _LossOnly(dict(DEFAULT_SYSTEM_CONFIG.loss))
    sysm.dataset, sysm.device, sysm.fused_batch = ds, torch.device(dev), True
    g = torch.Generator().manual_seed(4)
    n = 3001
    sysm.train_num_rays = n
    idx = torch.randint(0, 6, (n,), generator=g).to(dev)
    x = torch.randint(0, ds.w, (n,), generator=g).to(dev)
    y = torch.randint(0, ds.h, (n,), generator=g).to(dev)
    ref = sysm.preprocess_data_torch(idx, x, y)
    got = sysm.preprocess_data(idx, x, y)
    assert set(ref) == set(got)
    for k in ("rgb", "normal", "mask", "view_weights"):
        assert torch.equal(got[k], ref[k].reshape(got[k].shape)), k
    torch.testing.assert_close(got["rays"], ref["rays"], rtol=0, atol=1e-6)
    torch.testing.assert_close(got["cosines"], ref["cosines"], rtol=0, atol=1e-6)


def test_f16_table_image_follows_the_optimizer(dev):
    """Regression: torch's fused AdamW updates the master table without bumping its version
    counter; the f16 image the kernels read must still be rebuilt after every step."""
    ds = OrthoData.synthetic_sphere(128, device=dev)
    sysm = OrthoNeuSSystem(device=dev, seed=3)
    sysm.dataset = ds
    enc = sysm.model.geometry.hashgrid
    before = enc.params.detach().clone()
    for _ in range(3):
        sysm.training_step()
    assert not torch.equal(before, enc.params.detach())
    sysm.model.eval()
    assert torch.equal(enc.table_f16(), enc.params.detach().half())
    sysm.model.train()
    assert torch.equal(enc.table_f16(), enc.params.detach().half())


def test_table_adamw_matches_torch_adamw_with_progressive_levels(dev):
    """The fused hash-table optimizer (active levels only, lazy decay of the masked ones) against
    torch.optim.AdamW on the whole tensor with zero gradients on the masked levels: parameters,
    f16 image and the gradient reset, across a level switch and finalize()."""
    from drawingspinup_amd.nsr.encoding import Encoding
    from drawingspinup_amd.nsr.system import TableAdamW
    cfgd = {"otype": "HashGrid", "n_levels": 10, "n_features_per_level": 2, "log2_hashmap_size": 19,
            "base_resolution": 32, "per_level_scale": 1.3195079107728942}
    enc = Encoding(3, cfgd).to(dev).train()
    ref = enc.params.detach().clone().requires_grad_(True)
    topt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    opt = TableAdamW(enc, 1e-3, (0.9, 0.99), 1e-15)
    off = opt.offsets
    g = torch.Generator(device=dev).manual_seed(0)
    lrs = [1e-3, 1e-3, 8e-4, 6e-4, 5e-4, 4e-4, 3e-4]
    actives = [4, 4, 4, 5, 5, 7, 7]
    for lr, act in zip(lrs, actives):
        grad = torch.zeros_like(ref)
        grad[:off[act]] = torch.randn(off[act], device=dev, generator=g) * 1e-3
        grad[: off[1]] *= 50.0
        ref.grad = grad.clone()
        for pg in topt.param_groups:
            pg["lr"] = lr
        topt.step()
        opt.grad.copy_(grad)
        opt.step(act, lr)
        assert not opt.grad.any()                                    # gradient reset by the step
        n = off[act]
        torch.testing.assert_close(enc.params.detach()[:n], ref.detach()[:n], rtol=2e-6, atol=1e-9)
        assert torch.equal(enc.table_f16()[:n], enc.params.detach()[:n].half())
    opt.finalize()
    torch.testing.assert_close(enc.params.detach(), ref.detach(), rtol=5e-6, atol=1e-9)
    assert torch.equal(enc.table_f16(), enc.params.detach().half())
    # invalidation from outside (a checkpoint load) is honoured at the next step
    with torch.no_grad():
        enc.params.mul_(0.5)
        ref.mul_(0.5)
    enc.invalidate()
    ref.grad = torch.zeros_like(ref)
    topt.step()
    opt.step(7, lrs[-1])
    n = off[7]
    torch.testing.assert_close(enc.params.detach()[:n], ref.detach()[:n], rtol=5e-6, atol=1e-9)
    assert torch.equal(enc.table_f16()[:n], enc.params.detach()[:n].half())


def test_small_adamw_matches_torch_adamw(dev):
    """dsu_adamw_multi (one launch for all small tensors, per-group lr, per-tensor step count, tensors
    without a gradient skipped) against torch.optim.AdamW."""
    from drawingspinup_amd.nsr.system import SmallAdamW
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 23), (64,), (13, 64), (13,), (64, 16), (64, 64), (3, 64), (1,), (4096,)]
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]
    rs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    mk = lambda params: torch.optim.AdamW([{"params": params[:4], "lr": 1e-3}, {"params": params[4:7], "lr": 1e-2},
                                           {"params": params[7:], "lr": 5e-3}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    topt, holder = mk(rs), mk(ps)
    sopt = SmallAdamW(holder, (0.9, 0.99), 1e-15)
    for it in range(6):
        for grp_t, grp_s in zip(topt.param_groups, holder.param_groups):
            grp_t["lr"] = grp_s["lr"] = grp_s["lr"] * 0.9
        for k, (p, r) in enumerate(zip(ps, rs)):
            if k == 5 and it % 2 == 0:                   # a tensor that sometimes has no gradient
                p.grad = r.grad = None
                continue
            gr = (torch.randn(p.shape, generator=g) * (10.0 ** (k % 3 - 2))).to(dev)
            p.grad, r.grad = gr.clone(), gr.clone()
        topt.step()
        sopt.step()
        for p, r in zip(ps, rs):
            # absolute rounding differences scale with the step (lr <= 1e-2), not with the value
            torch.testing.assert_close(p.detach(), r.detach(), rtol=3e-6, atol=3e-8)


def test_side_stream_packing_matches_main_stream_packing(dev):
    """The prefetch path packs / appends the random points / sorts the next step's samples on the
    side stream into fixed-capacity buffers (sample total still on the device); the optimisation
    must follow the same trajectory as with the main-stream packing (same RNG draws; only the
    order inside a Morton bin, i.e. float summation order, may differ)."""
    def run(pack):
        ds = OrthoData.synthetic_sphere(256, device=dev)
        sysm = OrthoNeuSSystem(device=dev, seed=11)
        sysm.dataset = ds
        sysm.step_mode = "fused"                             # the Python-sequenced step is under test
        sysm.pack_on_side_stream = pack
        out = []
        for _ in range(40):
            r = sysm.training_step()
            out.append((float(r["loss"]), int(r["n_samples"]), int(r["n_rays"])))
        used = sum(1 for k in sysm._packed) if pack else 0
        return out, used
    a, used_a = run(True)
    b, used_b = run(False)
    assert used_a == 2 and used_b == 0                       # both parities of the packed buffers in use
    # INT: same rays, same sample counts (up to the first occupancy refresh at step 16, which
    # thresholds float densities)
    assert [x[1:] for x in a[:15]] == [x[1:] for x in b[:15]]
    la, lb = np.array([x[0] for x in a]), np.array([x[0] for x in b])
    np.testing.assert_allclose(la[:15], lb[:15], rtol=2e-3)
    assert la[-1] < la[0] and lb[-1] < lb[0]


def test_packed_step_buffers_equal_the_plain_calls(dev):
    """dsu_ray_compact_points_cap + dsu_points_tail + dsu_spatial_sort_dev (device-side total)
    against ray_march_finish + the tensor-level tail + dsu_spatial_sort."""
    g = torch.Generator().manual_seed(5)
    n = 700
    o = torch.cat([torch.rand(n, 2, generator=g) * 1.2 - 0.6, torch.full((n, 1), -2.0)], 1).to(dev)
    d = torch.tensor([[0.0, 0.0, 1.0]]).expand(n, 3).contiguous().to(dev)
    aabb = [-1.0] * 3 + [1.0] * 3
    tmin, tmax = ops.ray_aabb(o, d, aabb)
    occ = (torch.rand(128 ** 3, generator=g) < 0.3).to(torch.uint8).to(dev)
    step = 2 * 3 ** 0.5 / 1024
    h = ops.ray_march_begin(o, d, tmin, tmax, aabb, occ, 128, step)
    total, cmax = h.stats.tolist()
    pr = (torch.rand(64, 3, generator=g) * 2 - 1).to(dev)
    pe = torch.randn(64, 3, generator=g).to(dev)
    pts, ts, te = ops.ray_march_finish(h, total, cmax, tail_rows=128)
    pts[total:total + 64] = pr
    torch.add(pr, pe, alpha=1e-2, out=pts[total + 64:])
    for cap in (total + 1000, total, total - 37):             # roomy, exact, too small (rows dropped)
        bufs = ops.PackedStepBuffers(cap, 128, 6, dev)
        bufs.points.fill_(7.0)
        ops.ray_pack_prefetched(h, bufs, pr, pe, 1.0)
        m = min(total, cap)
        assert torch.equal(bufs.t_starts[:m], ts[:m]) and torch.equal(bufs.t_ends[:m], te[:m])
        assert torch.equal(bufs.points[:m], pts[:m])
        if cap >= total:
            nn = total + 128
            assert torch.equal(bufs.points[:total + 64], pts[:total + 64])
            # perturbed copies: torch.add(alpha=) may contract a + alpha * b into one fma
            torch.testing.assert_close(bufs.points[total + 64:nn], pts[total + 64:nn], rtol=0, atol=2e-7)
            perm = bufs.perm[:nn].long()
            assert torch.equal(torch.sort(perm).values, torch.arange(nn, device=dev))
            assert torch.equal(bufs.sorted[:nn], bufs.points[:nn][perm])


@pytest.mark.parametrize("n", [1, 63, 1000, 100003])
def test_shaded_texture_kernels_equal_the_separate_launches(dev, n):
    """dsu_texture_fwd_shaded / _bwd_shaded = shade_prep + texture MLP (+ their backward) in one
    launch each: the same arithmetic on the same values — per-sample results are bit-identical."""
    g = torch.Generator().manual_seed(n)
    mk = lambda *s: (torch.randn(*s, generator=g)).to(dev)
    params = [mk(64, 16) * 0.3, mk(64) * 0.1, mk(64, 64) * 0.15, mk(64) * 0.1, mk(3, 64) * 0.2,
              mk(3) * 0.1]
    feat, grad = mk(n + 7, 13), mk(n + 7, 3)
    grad[0] = 0.0                                            # zero gradient: the clamped-norm branch
    d_rgb, d_normal = mk(n, 3), mk(n, 3)
    normal, tex_in = ops.shade_prep_fwd(grad[:n].contiguous(), feat[:n].contiguous())
    rgb = ops.texture_fwd(params, tex_in)
    d_tex_in, gp = ops.texture_bwd(params, tex_in, rgb, d_rgb)
    d_grad, d_feat = ops.shade_prep_bwd(grad[:n].contiguous(), d_normal, d_tex_in)
    normal2, rgb2 = ops.texture_fwd_shaded(params, feat[:n].contiguous(), grad[:n].contiguous())
    assert torch.equal(normal2, normal) and torch.equal(rgb2, rgb)
    d_grad2, d_feat2, gp2 = ops.texture_bwd_shaded(params, feat[:n].contiguous(), grad[:n].contiguous(),
                                                   rgb2, d_rgb, d_normal, tail_rows=7)
    assert torch.equal(d_grad2, d_grad)
    assert torch.equal(d_feat2[:n], d_feat) and float(d_feat2[n:].abs().max()) == 0.0
    for a, b in zip(gp, gp2):          # sums over the samples: the two instantiations agree to rounding
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5 * float(a.abs().max()))


@pytest.mark.parametrize("n", [63, 5000, 100003])
def test_deferred_partial_sums_in_the_scatter_launch_equal_the_separate_reductions(dev, n):
    """The native step's tail: the texture backward leaves its per-workgroup partial gradients and
    the geometry backward's scatter launch sums them (and its own) in its first workgroups
    (dsu_texture_bwd_shaded_partials + dsu_sdf_fd_bwd_sorted_fold).  Same partials, same summation
    order as the stand-alone reduction launches: the geometry gradients are bit-identical."""
    g = torch.Generator().manual_seed(n)
    mk = lambda *s: (torch.randn(*s, generator=g)).to(dev)
    params = [mk(64, 16) * 0.3, mk(64) * 0.1, mk(64, 64) * 0.15, mk(64) * 0.1, mk(3, 64) * 0.2,
              mk(3) * 0.1]
    feat, grad = mk(n, 13), mk(n, 3)
    d_rgb, d_normal = mk(n, 3), mk(n, 3)
    _, rgb = ops.texture_fwd_shaded(params, feat, grad)
    d_grad, d_feat, gp = ops.texture_bwd_shaded(params, feat, grad, rgb, d_rgb, d_normal, tail_rows=3)
    d_grad2, d_feat2, flat, deferred = ops.texture_bwd_shaded_partials(params, feat, grad, rgb, d_rgb,
                                                                      d_normal, tail_rows=3)
    assert torch.equal(d_grad2, d_grad) and torch.equal(d_feat2, d_feat)
    assert float(flat.abs().max()) == 0.0                      # nothing summed yet
    # a geometry backward on the same stream, with and without the deferred record
    cfg = ops.HashGridConfig()
    tab = ((torch.rand(cfg.n_entries, 2, generator=g) * 2 - 1) * 0.1).half().to(dev)
    mlp = [mk(64, 23) * 0.3, mk(64) * 0.05, mk(13, 64) * 0.2, mk(13) * 0.1]
    pts = (torch.rand(n, 3, generator=g) * 1.6 - 0.8).to(dev)
    d = [mk(n), mk(n, 3), mk(n, 13), None]
    gt_a, g_a = ops.sdf_fd_bwd(cfg, tab, mlp, pts, 1.0, 0.01, 5, *d)
    gt_b, g_b = ops.sdf_fd_bwd(cfg, tab, mlp, pts, 1.0, 0.01, 5, *d, extra=deferred)
    for a, b in zip(g_a, g_b):
        assert torch.equal(a, b)
    torch.testing.assert_close(gt_b, gt_a, rtol=1e-4, atol=1e-5 * float(gt_a.abs().max()))   # atomics
    sizes = [t.numel() for t in params]
    # (two runs of the texture backward: its waves add into the workgroup's partial vector with LDS
    # float atomics, so the partials themselves differ in the last bits from run to run)
    for a, b in zip(gp, torch.split(flat, sizes)):
        torch.testing.assert_close(b, a.reshape(-1), rtol=1e-5, atol=1e-5 * float(a.abs().max()))
