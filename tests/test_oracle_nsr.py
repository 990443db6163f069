"""CPU: the NSR oracle and the host-side mirror against outputs of the reference's own
instant_nsr modules (tests/golden/nsr_reference.npz, made by tests/golden/make_nsr_golden.py),
plus hand-derived known-answer tests for the restated third-party ops."""
import os

import numpy as np
import torch

from oracle import hashgrid as oh
from oracle import nerfacc_ref as nr

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "nsr_reference.npz"))
LV = oh.make_levels()


def gold_table():
    g = torch.Generator().manual_seed(int(GOLD["table_seed"]))
    n = LV["offsets"][10] * 2
    p = (torch.rand(n, generator=g) * 2 - 1) * float(GOLD["table_scale"])
    return p.half().numpy().reshape(-1, 2)


def gold_mlp():
    return [GOLD["w0"], GOLD["b0"], GOLD["w1"], GOLD["b1"]]


# ------------------------------------------------------------------ golden: reference modules
def test_oracle_sdf_fd_matches_reference_volume_sdf():
    tab, mlp = gold_table(), gold_mlp()
    for step, level in ((0, 4), (1500, 5), (2999, 6)):
        k = f"s{step}."
        assert int(GOLD[k + "level"]) == level
        eps = float(GOLD[k + "eps"])
        sdf, grad, feat, lap = oh.sdf_fd(tab, mlp, GOLD[k + "pts"], 1.0, eps, LV, level)
        np.testing.assert_allclose(sdf, GOLD[k + "sdf"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(feat, GOLD[k + "feature"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(sdf, GOLD[k + "forward_level"], rtol=0, atol=2e-6)
        # the reference differences f32 network outputs: |err| <= ~2 ulp(sdf) / eps
        np.testing.assert_allclose(grad, GOLD[k + "grad"], rtol=0, atol=4e-7 / eps)
        np.testing.assert_allclose(lap, GOLD[k + "laplace"], rtol=0, atol=6e-6 / eps ** 2)


def test_host_schedule_and_keys_match_reference():
    from drawingspinup_amd.nsr.model import Cfg, DEFAULT_MODEL_CONFIG, NeuSModel
    cfg = Cfg(DEFAULT_MODEL_CONFIG)
    cfg["grid_prune"] = False
    m = NeuSModel(cfg)
    ref_keys = [str(k) for k in GOLD["state_dict_keys"]]
    assert sorted(m.state_dict().keys()) == sorted(ref_keys)
    m.train()
    for step in (0, 1500, 2999):
        m.update_step(0, step)
        k = f"s{step}."
        assert m.geometry.active_levels == int(GOLD[k + "level"])
        assert m.geometry._finite_difference_eps == float(GOLD[k + "eps"])     # bit-exact
        assert m.cos_anneal_ratio == float(GOLD[k + "cos_anneal"])
    assert m.render_step_size == float(GOLD["render_step_size"])
    # weight-norm: effective weight of layer 0 from (g, v)
    w = torch._weight_norm(torch.from_numpy(GOLD["w0_v"]), torch.from_numpy(GOLD["w0_g"]), 0)
    np.testing.assert_allclose(w.numpy(), GOLD["w0"], rtol=1e-6, atol=1e-7)


def test_host_alpha_texture_losses_match_reference():
    from drawingspinup_amd.nsr.model import Cfg, DEFAULT_MODEL_CONFIG, NeuSModel
    from drawingspinup_amd.nsr.system import binary_cross_entropy, ranking_loss
    cfg = Cfg(DEFAULT_MODEL_CONFIG)
    cfg["grid_prune"] = False
    m = NeuSModel(cfg)
    with torch.no_grad():
        m.variance.variance.copy_(torch.from_numpy(GOLD["variance"]))
        for i in (0, 2, 4):
            m.texture.network.layers[i].weight.copy_(torch.from_numpy(GOLD[f"tex.w{i}"]))
            m.texture.network.layers[i].bias.copy_(torch.from_numpy(GOLD[f"tex.b{i}"]))
    m.train()
    for step in (0, 1500, 2999):
        m.update_step(0, step)
        k = f"s{step}."
        sdf, grad = torch.from_numpy(GOLD[k + "sdf"]), torch.from_numpy(GOLD[k + "grad"])
        dirs = torch.from_numpy(GOLD[k + "dirs"])
        normal = torch.nn.functional.normalize(grad, dim=-1)
        dists = torch.full((sdf.shape[0], 1), m.render_step_size)
        alpha = m.get_alpha(sdf, normal, dirs, dists)
        np.testing.assert_allclose(alpha.detach().numpy(), GOLD[k + "alpha"], rtol=1e-5, atol=1e-6)
        rgb = m.texture(torch.from_numpy(GOLD[k + "feature"]), dirs, normal)
        np.testing.assert_allclose(rgb.detach().numpy(), GOLD[k + "rgb"], rtol=1e-5, atol=1e-6)
    err, w = torch.from_numpy(GOLD["rank.err"]), torch.from_numpy(GOLD["rank.w"])
    assert float(ranking_loss(err, 0.8, None, "mean")) == float(GOLD["rank.mean08"])
    np.testing.assert_allclose(float(ranking_loss(err, 0.9, w, "sum")), float(GOLD["rank.sum09w"]),
                               rtol=1e-6)
    bce = binary_cross_entropy(err.clamp(1e-3, 1 - 1e-3), (w > 0.5).float())
    np.testing.assert_allclose(bce.numpy(), GOLD["bce"], rtol=1e-6)


# ------------------------------------------------------------------ KATs: restated tcnn grid
def test_level_table_kats():
    # res 32/43/56/74 dense, then hashed; 3 838 848 entries in total
    assert LV["resolution"][:4] == [32, 43, 56, 74]
    assert LV["offsets"][:5] == [0, 32768, 112280, 287896, 693120]
    assert LV["offsets"][10] == 3838848
    assert LV["hashed"] == [0, 0, 0, 0, 1, 1, 1, 1, 1, 1]
    assert float(LV["scale"][0]) == 31.0 and float(LV["scale"][5]) == 127.0


def test_index_kats():
    one = lambda v: np.array([v], np.uint32)
    # hashed level: (1*1) ^ (2*2654435761) ^ (3*805459861) mod 2^19 in uint32 arithmetic
    exp = (1 ^ ((2 * 2654435761) & 0xFFFFFFFF) ^ ((3 * 805459861) & 0xFFFFFFFF)) % (1 << 19)
    assert int(oh._index(LV, 6, one(1), one(2), one(3))[0]) == exp
    # dense level: x + y*res + z*res^2
    assert int(oh._index(LV, 1, one(5), one(6), one(7))[0]) == 5 + 6 * 43 + 7 * 43 * 43
    # x == 1.0 on a dense level steps to cell res-1 with corner res: wraps modulo the level size
    idx, w = oh.corner_indices_weights(LV, 0, np.array([[1.0, 1.0, 1.0]], np.float32))
    assert idx.max() < 32768 and abs(float(w.sum()) - 1.0) < 1e-6


def test_encode_kats():
    rng = np.random.default_rng(0)
    tab = (rng.random((LV["offsets"][10], 2)).astype(np.float32) - 0.5).astype(np.float16)
    # at an exact lattice vertex of level 0 the encoding is the table entry itself
    x = np.array([[(5 - 0.5) / 31.0, (6 - 0.5) / 31.0, (7 - 0.5) / 31.0]], np.float64)
    x = x.astype(np.float32)
    enc = oh.encode(tab, x, LV, 1)
    cell, frac = oh._cell(LV["scale"][0], x)
    if np.all(frac == 0):
        assert np.array_equal(enc[0, :2], tab[5 + 6 * 32 + 7 * 1024])
    # masked levels are exactly zero; constant table -> constant features (weights sum to 1)
    assert not oh.encode(tab, rng.random((50, 3)).astype(np.float32), LV, 3)[:, 6:].any()
    const = np.full_like(tab, np.float16(0.25))
    e = oh.encode(const, rng.random((200, 3)).astype(np.float32), LV, 10).astype(np.float32)
    assert np.abs(e - 0.25).max() < 2e-3
    # gradient of sum(enc) w.r.t. the table sums to (number of points) per active feature
    g = oh.encode_bwd(rng.random((64, 3)).astype(np.float32), np.ones((64, 20)), LV, 2)
    np.testing.assert_allclose(g[:LV["offsets"][2]].sum(0), [128.0, 128.0], rtol=1e-6)
    assert not g[LV["offsets"][2]:].any()


# ------------------------------------------------------------------ KATs: restated nerfacc ops
def test_nerfacc_kats():
    aabb = [-1, -1, -1, 1, 1, 1]
    o = np.array([[0.1, 0.2, -1.3], [3.0, 0.0, -1.3]], np.float32)
    d = np.array([[0, 0, 1.0], [0, 0, 1.0]], np.float32)
    tn, tf = nr.ray_aabb_intersect(o, d, aabb)
    assert abs(tn[0] - 0.3) < 1e-6 and abs(tf[0] - 2.3) < 1e-6
    assert tn[1] == np.float32(1e10) and tf[1] == np.float32(1e10)      # miss
    step = 1.732 * 2 / 1024
    empty = np.zeros(128 ** 3, np.uint8)
    ri, ts, te, cnt = nr.ray_marching(o[:1], d[:1], tn[:1], tf[:1], aabb, empty, 128, step)
    assert len(ri) == 0 and cnt[0] == 0                                # empty grid: 0 samples
    ri, ts, te, cnt = nr.ray_marching(o[:1], d[:1], tn[:1], tf[:1], aabb, None, 0, step)
    assert abs(int(cnt[0]) - int(2.0 / step)) <= 1                     # dense: (far-near)/step
    assert np.allclose(te - ts, step, atol=1e-6) and np.all(np.diff(ts) > 0)
    # a slab of occupied cells (z in [0, 0.25)): samples only inside the slab
    occ = np.zeros((128, 128, 128), np.uint8)
    occ[:, :, 64:80] = 1
    ri, ts, te, cnt = nr.ray_marching(o[:1], d[:1], tn[:1], tf[:1], aabb, occ.reshape(-1), 128, step)
    mid = (ts + te) / 2 - 1.3
    assert cnt[0] > 0 and mid.min() >= 0.0 - 1e-6 and mid.max() < 0.25 + 1e-6
    assert abs(int(cnt[0]) - int(0.25 / step)) <= 1
    # weights: alpha == 1 on the first sample -> one-hot
    w = nr.render_weight_from_alpha(np.array([1.0, 0.3, 0.7]), [3])
    assert w.tolist() == [1.0, 0.0, 0.0]
    w = nr.render_weight_from_alpha(np.array([0.5, 0.5, 0.5, 0.2]), [3, 1])
    np.testing.assert_allclose(w, [0.5, 0.25, 0.125, 0.2])
    acc = nr.accumulate_along_rays(w, None, np.array([0, 0, 0, 1]), 3)
    np.testing.assert_allclose(acc[:, 0], [0.875, 0.2, 0.0])
    # occupancy EMA + mean-clamped threshold
    occs, binary = nr.occgrid_update(np.array([0.0, 0.02, 0.0, 0.0], np.float32), None,
                                     np.array([0.0, 0.0, 0.004, 0.0], np.float32), 0.95, 0.001)
    np.testing.assert_allclose(occs, [0.0, 0.019, 0.004, 0.0], rtol=1e-6)
    assert binary.tolist() == [False, True, True, False]


def test_sync_free_ranking_loss_equals_reference_form():
    """ranking_loss_masked(e, m, ...) == ranking_loss(e[m], ..., w[m]) (criterions.py:16-27)."""
    from drawingspinup_amd.nsr.system import ranking_loss, ranking_loss_masked
    g = torch.Generator().manual_seed(3)
    for n in (1, 7, 10, 100, 1000):
        for ratio in (0.8, 0.9, 0.7):
            e = torch.rand(n, generator=g, dtype=torch.float32)
            w = torch.rand(n, generator=g)
            m = torch.rand(n, generator=g) > 0.3
            if int(m.sum()) == 0 or int(ratio * int(m.sum())) == 0:
                continue
            for typ in ("mean", "sum"):
                a = ranking_loss_masked(e, m, ratio, w, typ)
                b = ranking_loss(e[m], ratio, w[m], typ)
                torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)
            torch.testing.assert_close(ranking_loss_masked(e, torch.ones_like(m), ratio),
                                       ranking_loss(e, ratio), rtol=1e-6, atol=1e-7)
    # gradients agree with the reference form too (it selects sorted[indices[:k]], see docstring)
    e1 = torch.tensor([0.5, 0.1, 0.9, 0.3, 0.7], requires_grad=True)
    e2 = e1.detach().clone().requires_grad_(True)
    m = torch.tensor([True, True, False, True, True])
    ranking_loss_masked(e1, m, 0.7, None, "mean").backward()
    ranking_loss(e2[m], 0.7, None, "mean").backward()
    assert torch.equal(e1.grad, e2.grad)


def test_torch_cpu_form_of_the_geometry_pass_agrees_with_the_numpy_oracle():
    """oracle/hashgrid_torch.py (bench.py's multi-threaded cpu_baseline leg) computes what the
    bit-faithful numpy oracle computes, to f16 rounding of the interpolated features."""
    import torch
    from oracle import hashgrid_torch as ht
    lv = oh.make_levels()
    g = torch.Generator().manual_seed(0)
    tab = ((torch.rand(int(lv["offsets"][10]), 2, generator=g) * 2 - 1) * 0.1).half()
    mlp = [torch.randn(*s, generator=g) * 0.2 for s in [(64, 23), (64,), (13, 64), (13,)]]
    pts = torch.rand(300, 3, generator=g) * 2 - 1
    for active in (4, 7, 10):
        a = ht.sdf_fd(tab.float(), mlp, pts, 1.0, 0.02, lv, active)
        b = oh.sdf_fd(tab.numpy(), [m.numpy() for m in mlp], pts.numpy(), 1.0, 0.02, lv, active)
        np.testing.assert_allclose(a[0].numpy(), b[0], rtol=0, atol=2e-4)
        np.testing.assert_allclose(a[2].numpy(), b[2], rtol=0, atol=3e-4)
        np.testing.assert_allclose(a[1].numpy(), b[1], rtol=0, atol=2e-4 / 0.02)
    fwd, both = ht.training_work_seconds(2000, 5)
    assert 0 < fwd < both
