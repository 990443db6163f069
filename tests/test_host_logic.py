"""CPU: host-side mirrors of the reference's glue (scheduler, camera embedding, sharding)."""
import numpy as np
import torch

from drawingspinup_amd.mv.pipeline import DDIMScheduler, DEFAULT_CAMERA_EMBEDDING
from drawingspinup_amd.nsr import system as S


def test_ddim_timesteps_and_step():
    s = DDIMScheduler()
    s.set_timesteps(75)
    ts = s.timesteps.tolist()
    # 'leading' spacing with steps_offset=1: step ratio 1000//75 = 13 -> 962+1, ..., 0+1
    assert len(ts) == 75 and ts[0] == 74 * 13 + 1 and ts[-1] == 1 and ts[0] - ts[1] == 13
    # alphas_cumprod of the scaled-linear schedule (SD-1.x): known end points
    assert abs(float(s.alphas_cumprod[0]) - (1 - 0.00085)) < 1e-7
    assert abs(float(s.alphas_cumprod[-1]) - 0.0047) < 2e-4
    # eta = 0, exact epsilon: DDIM inverts q(x_t | x_0) deterministically
    g = torch.Generator().manual_seed(0)
    x0, eps = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    t = ts[10]
    a = s.alphas_cumprod[t]
    xt = a ** 0.5 * x0 + (1 - a) ** 0.5 * eps
    prev = s.step(eps, t, xt, eta=0.0)
    ap = s.alphas_cumprod[t - 13]
    torch.testing.assert_close(prev, ap ** 0.5 * x0 + (1 - ap) ** 0.5 * eps, rtol=1e-4, atol=1e-4)
    # eta = 1 uses the injected variance noise with sigma_t
    noise = torch.randn(2, 4, 8, 8, generator=g)
    p1 = s.step(eps, t, xt, eta=1.0, variance_noise=noise)
    var = ((1 - ap) / (1 - a)) * (1 - a / ap)
    exp = ap ** 0.5 * x0 + (1 - ap - var) ** 0.5 * eps + var ** 0.5 * noise
    torch.testing.assert_close(p1, exp, rtol=1e-4, atol=1e-4)


def test_camera_embedding_table():
    # rows equal SingleImageDataset.get_T of the nine_views poses (f16 rounded)
    ce = DEFAULT_CAMERA_EMBEDDING
    assert ce.shape == (12, 5)
    np.testing.assert_allclose(ce[1, 1:3].float().numpy(), [-0.23624, 0.81238], atol=6e-4)
    np.testing.assert_allclose(ce[3, 1:3].float().numpy(), [0.52204, 3.14159], atol=2e-3)
    assert ce[:6, 3].tolist() == [1.0] * 6 and ce[6:, 4].tolist() == [1.0] * 6   # task one-hots


def test_pose_generator_matches_fixed_pose_structure():
    # instant_nsr/datasets/fixed_poses/000_front_RT.txt: rows (1,0,0|0), (0,0,1|0), (0,-1,0|-1.3)
    f = S.ideal_w2c("front")
    np.testing.assert_allclose(f, [[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, -1.3]], atol=1e-12)
    r = S.ideal_w2c("right")
    np.testing.assert_allclose(r, [[0, 1, 0, 0], [0, 0, 1, 0], [1, 0, 0, -1.3]], atol=1e-12)
    fl = S.ideal_w2c("front_left")
    np.testing.assert_allclose(fl[2], [-0.70710678, -0.70710678, 0, -1.83847763], atol=1e-6)
    # ortho rays (ray_utils.py:20-33): pixel centres, origin plane z=0, direction +z
    o, d = S.ortho_rays_hw(4, 2)
    assert o.shape == (2, 4, 3) and torch.equal(d[..., 2], torch.ones(2, 4))
    np.testing.assert_allclose(o[0, :, 0].numpy(), [-0.75, -0.25, 0.25, 0.75])
    np.testing.assert_allclose(o[:, 0, 1].numpy(), [-0.5, 0.5])


def test_lr_schedule_and_param_groups():
    """configs/neuralangelo-ortho-wmask.yaml:101-127: AdamW groups geometry 1e-3 / texture 1e-2 /
    variance 1e-3, constant for 500 steps then exponential decay to 0.1x at step 3000."""
    sysm = S.OrthoNeuSSystem(device="cpu")
    names = [g["name"] for g in sysm.optimizer.param_groups]
    assert names == ["geometry", "texture", "variance"]
    n_geo = sum(p.numel() for p in sysm.optimizer.param_groups[0]["params"])
    assert n_geo == 3838848 * 2 + 64 * 23 + 64 + 64 + 13 * 64 + 13 + 13      # table + weight-normed MLP
    for step, factor in ((0, 1.0), (499, 1.0), (500, 1.0), (1750, 0.1 ** 0.5), (3000, 0.1)):
        sysm.global_step = step
        sysm._set_lr()
        lrs = [g["lr"] for g in sysm.optimizer.param_groups]
        np.testing.assert_allclose(lrs, [1e-3 * factor, 1e-2 * factor, 1e-3 * factor], rtol=1e-9)
    assert sysm.train_num_rays == 256 and sysm.train_num_samples == 256 * 1024


def test_sharding():
    from drawingspinup_amd import dist
    uids = list(range(24))
    parts = [dist.shard(uids, r, 8) for r in range(8)]
    assert sorted(sum(parts, [])) == uids and all(len(p) == 3 for p in parts)
    assert dist.shard(uids, 0, 1) == uids


# ---------------------------------------------------------------------------------------------
# Host bookkeeping of the fused optimizers (the kernels are stood in for by torch CPU code — test
# infrastructure only; the GPU tests run the real kernels against torch.optim.AdamW)
# ---------------------------------------------------------------------------------------------
def _cpu_adamw(p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2s):
    import torch
    P, G, M, V = p[:n], g[:n], m[:n], v[:n]
    P.mul_(1 - lr * wd)
    M.lerp_(G, 1 - b1)
    V.mul_(b2).addcmul_(G, G, value=1 - b2)
    P.addcdiv_(M, V.sqrt() / bc2s + eps, value=-lr / bc1)
    G.zero_()


def test_table_adamw_lazy_decay_bookkeeping(monkeypatch):
    """TableAdamW: levels the schedule still masks only accumulate their decay factor on the host;
    it is applied when a level is switched on and in finalize().  Against torch.optim.AdamW on the
    whole tensor (zero gradients on the masked levels), on the CPU with stub kernels."""
    import torch
    from drawingspinup_amd import ops
    from drawingspinup_amd.nsr import system as S
    from drawingspinup_amd.nsr.encoding import Encoding

    def table_adamw(p, g, m, v, img, n, lr, b1, b2, eps, wd, bc1, bc2s):
        _cpu_adamw(p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2s)
        img[:n] = p[:n].half()

    def table_decay(p, img, start, n, factor):
        p[start:start + n].mul_(factor)
        img[start:start + n] = p[start:start + n].half()

    monkeypatch.setattr(ops, "table_adamw", table_adamw)
    monkeypatch.setattr(ops, "table_decay", table_decay)
    enc = Encoding(3, {"otype": "HashGrid", "n_levels": 10, "n_features_per_level": 2,
                       "log2_hashmap_size": 12, "base_resolution": 4,
                       "per_level_scale": 1.3195079107728942}).train()
    ref = enc.params.detach().clone().double().requires_grad_(True)
    topt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    opt = S.TableAdamW(enc, 1e-3, (0.9, 0.99), 1e-15)
    off = opt.offsets
    g = torch.Generator().manual_seed(0)
    for it, (lr, act) in enumerate(zip([1e-3, 9e-4, 8e-4, 7e-4, 6e-4, 5e-4], [4, 4, 6, 6, 6, 9])):
        grad = torch.zeros(ref.shape[0])
        grad[:off[act]] = torch.randn(off[act], generator=g) * 1e-3
        ref.grad = grad.double()
        topt.param_groups[0]["lr"] = lr
        topt.step()
        opt.grad.copy_(grad)
        opt.step(act, lr)
        assert opt.active == act and not opt.grad.any()
        n = off[act]
        torch.testing.assert_close(enc.params.detach()[:n].double(), ref.detach()[:n], rtol=1e-5, atol=1e-8)
        if act < 10:                                          # masked levels untouched so far
            assert opt.pending < 1.0
    opt.finalize()
    assert opt.pending == 1.0 and opt.active == 9
    torch.testing.assert_close(enc.params.detach().double(), ref.detach(), rtol=1e-5, atol=1e-8)
    assert torch.equal(enc.table_f16(), enc.params.detach().half())      # image kept in step
    enc.invalidate()                                                     # e.g. a checkpoint load
    assert not enc._shadow_locked


def test_small_adamw_entries_and_step_counts(monkeypatch):
    import math
    import torch
    from drawingspinup_amd import ops
    from drawingspinup_amd.nsr import system as S
    seen = []

    def adamw_multi(entries, b1, b2, eps, wd):
        seen.append(len(entries))
        for p, g, m, v, lr, bc1, bc2s in entries:
            _cpu_adamw(p.view(-1), g.clone().view(-1), m.view(-1), v.view(-1), p.numel(), lr, b1, b2, eps, wd,
                       bc1, bc2s)

    monkeypatch.setattr(ops, "adamw_multi", adamw_multi)
    ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5)),
          torch.nn.Parameter(torch.randn(1))]
    rs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    mk = lambda q: torch.optim.AdamW([{"params": q[:2], "lr": 1e-3}, {"params": q[2:], "lr": 1e-2}],
                                     lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    holder, topt = mk(ps), mk(rs)
    sopt = S.SmallAdamW(holder, (0.9, 0.99), 1e-15)
    for it in range(4):
        for k, (p, r) in enumerate(zip(ps, rs)):
            if k == 1 and it == 1:
                p.grad = r.grad = None                        # skipped, step count not advanced
                continue
            gr = torch.randn(p.shape)
            p.grad, r.grad = gr.clone(), gr.clone()
        topt.step()
        sopt.step()
        for p, r in zip(ps, rs):
            torch.testing.assert_close(p.detach(), r.detach(), rtol=1e-5, atol=1e-8)
    assert seen == [3, 2, 3, 3]
    assert [sopt.state[id(p)][2] for p in ps] == [4, 3, 4]
    assert math.isclose(topt.state[rs[1]]["step"].item(), 3)


def test_level_outer_forward_preconditions_of_the_shipped_grid():
    """What sdf_fd_fwd_shared_kernel (csrc/hashgrid.hip) relies on for REGULAR points, checked with
    the oracle's level table and the reference's progressive eps (geometry.py:196-215):
    levels 0-3 dense / 4-9 hashed (its ND = 4); a +-eps offset moves less than one cell on every
    active level (so the offset's cell is the centre's or its face neighbour); cell coordinates of
    points in [0, 1] stay in [0, res - 1], hence a dense corner index is below 2 x the level size
    (one subtraction replaces tcnn's modulo)."""
    import numpy as np
    from oracle import hashgrid as oh
    lv = oh.make_levels()
    assert lv["hashed"] == [0, 0, 0, 0, 1, 1, 1, 1, 1, 1]
    f32 = np.float32
    rng = np.random.default_rng(0)
    p = (rng.random((200000, 3), dtype=np.float32) * 2 - 1).astype(f32)
    p[:6] = [[1, -1, 0.3], [-1, 1, 1], [0.99999, 0, -0.99999], [1, 1, 1], [-1, -1, -1], [0, 0, 0]]
    radius = f32(1.0)
    contract = lambda x: ((x - (-radius)) / (radius - (-radius))).astype(f32)
    for active in range(1, 11):
        eps = f32(2.0 * 1.0 / (32 * 1.3195079107728942 ** (active - 1)))
        q0 = contract(p)
        for l in range(active):
            scale, res = f32(lv["scale"][l]), lv["resolution"][l]
            size = lv["offsets"][l + 1] - lv["offsets"][l]
            assert float(eps) / 2 * float(scale) < 1.0
            c0 = np.floor((np.float64(scale) * q0.astype(np.float64) + 0.5).astype(f32))
            assert c0.min() >= 0 and c0.max() <= res - 1
            if not lv["hashed"][l]:
                assert res + res * res + res ** 3 < 2 * size      # largest corner index (coordinate res)
            for ax in range(3):
                for sgn in (1.0, -1.0):
                    qe = contract(np.clip(p[:, ax] + f32(sgn) * eps, -radius, radius).astype(f32))
                    assert qe.min() >= 0.0 and qe.max() <= 1.0
                    ce = np.floor((np.float64(scale) * qe.astype(np.float64) + 0.5).astype(f32))
                    rel = ce - c0[:, ax]
                    assert rel.min() >= -1 and rel.max() <= 1, (active, l, ax, sgn)
