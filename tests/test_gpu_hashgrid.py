"""HIP hash-grid / fused SDF kernels vs the CPU oracle (through the C ABI)."""
import numpy as np
import pytest
import torch

from drawingspinup_amd import ops
from oracle import hashgrid as oh

pytestmark = pytest.mark.gpu

CFG = ops.HashGridConfig()
LV = oh.make_levels()


def _table(seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(CFG.n_entries, 2, generator=g) * 2 - 1) * scale).half()


def _mlp(seed, din=23):
    g = torch.Generator().manual_seed(seed)
    w0 = torch.randn(64, din, generator=g) * 0.3
    b0 = torch.randn(64, generator=g) * 0.05
    w1 = torch.randn(13, 64, generator=g) * 0.2
    b1 = torch.randn(13, generator=g) * 0.1
    return [w0, b0, w1, b1]


def _pts(n, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, 3, generator=g) * (hi - lo) + lo


@pytest.mark.parametrize("active", [0, 4, 7, 10])
def test_encode_fwd_bit_exact(dev, active):
    tab = _table(1)
    x = _pts(4099, 2)
    # exercise the domain edges too
    x[0] = 0.0
    x[1] = 1.0
    x[2] = torch.tensor([1.0, 0.0, 0.5])
    ref = oh.encode(tab.numpy(), x.numpy(), LV, active)
    out = ops.hashgrid_encode_fwd(CFG, tab.to(dev), x.to(dev), active).cpu().numpy()
    assert out.dtype == np.float16 and out.shape == (4099, 20)
    # f16 FMA chain restated exactly -> bit-exact
    assert np.array_equal(out.view(np.uint16), ref.view(np.uint16))


def test_encode_empty(dev):
    tab = _table(1).to(dev)
    out = ops.hashgrid_encode_fwd(CFG, tab, torch.empty(0, 3, device=dev), 4)
    assert out.shape == (0, 20)


def test_encode_bwd(dev):
    x = _pts(3001, 3)
    g = torch.Generator().manual_seed(4)
    dout = torch.randn(3001, 20, generator=g)
    active = 6
    ref = oh.encode_bwd(x.numpy(), dout.numpy().astype(np.float64), LV, active).reshape(-1)
    got = ops.hashgrid_encode_bwd(CFG, x.to(dev), dout.to(dev), active).cpu().numpy()
    # same set of touched entries (integer indexing) ...
    assert np.array_equal(ref != 0, got != 0)
    # ... and values to f32 atomic-order rounding
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5)
    # masked levels receive nothing
    assert not got[LV["offsets"][active] * 2:].any()


@pytest.mark.parametrize("active,n_out", [(4, 1), (4, 13), (10, 13), (6, 1)])
def test_sdf_fwd(dev, active, n_out):
    tab = _table(5, 0.5)
    mlp = _mlp(6)
    pts = _pts(5000, 7, -1.0, 1.0)
    ref = oh.sdf_network(tab.numpy(), [m.numpy() for m in mlp], pts.numpy(), 1.0, LV, active)
    out = ops.sdf_fwd(CFG, tab.to(dev), [m.to(dev) for m in mlp], pts.to(dev), 1.0, active,
                      n_out).cpu().numpy()
    np.testing.assert_allclose(out, ref[:, :n_out], rtol=2e-5, atol=2e-5)


def test_sdf_fd_fwd(dev):
    tab = _table(8, 0.5)
    mlp = _mlp(9)
    pts = _pts(3000, 10, -1.0, 1.0)
    pts[0] = torch.tensor([1.0, -1.0, 0.999])   # clamp path of the +-eps offsets
    eps, active = 0.0213, 5
    sdf, grad, feat, lap = oh.sdf_fd(tab.numpy(), [m.numpy() for m in mlp], pts.numpy(), 1.0,
                                     eps, LV, active)
    o = ops.sdf_fd_fwd(CFG, tab.to(dev), [m.to(dev) for m in mlp], pts.to(dev), 1.0, eps, active)
    np.testing.assert_allclose(o[0].cpu().numpy(), sdf, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(o[2].cpu().numpy(), feat, rtol=2e-5, atol=2e-5)
    # differences of f32 sdf values divided by eps / eps^2 amplify rounding
    np.testing.assert_allclose(o[1].cpu().numpy(), grad, rtol=1e-3, atol=2e-5 / eps)
    np.testing.assert_allclose(o[3].cpu().numpy(), lap, rtol=1e-3, atol=8e-5 / eps ** 2)


def _torch_fd_loss(tab64, mlp64, pts, eps, active, radius, d):
    """Differentiable float64 restatement (indices/weights from the oracle) for gradients.
    Forward VALUES of the features are the f16-rounded ones the real op produces (tcnn
    semantics); the gradient is that of the un-rounded trilinear blend (straight-through)."""
    tab16 = tab64.detach().numpy().astype(np.float16)

    def net(p):
        xc = oh.contract(p, radius)
        feats = []
        enc16 = torch.from_numpy(oh.encode(tab16, xc, LV, active).astype(np.float64))
        for l in range(active):
            idx, w = oh.corner_indices_weights(LV, l, xc)
            t = tab64[LV["offsets"][l]:LV["offsets"][l + 1]]
            f = (t[torch.from_numpy(idx)] * torch.from_numpy(w).double().unsqueeze(-1)).sum(1)
            feats.append(f + (enc16[:, 2 * l:2 * l + 2] - f).detach())
        nl = len(LV["resolution"])
        feats.append(torch.zeros(p.shape[0], 2 * (nl - active), dtype=torch.float64))
        xyz = torch.from_numpy(xc.astype(np.float32) * np.float32(2) + np.float32(-1)).double()
        inp = torch.cat([xyz] + feats, 1)
        h = torch.nn.functional.softplus(inp @ mlp64[0].T + mlp64[1], beta=100)
        return h @ mlp64[2].T + mlp64[3]
    out = net(pts)
    sdf = out[:, 0]
    e = np.float32(eps)
    offs = np.array([[e, 0, 0], [-e, 0, 0], [0, e, 0], [0, -e, 0], [0, 0, e], [0, 0, -e]],
                    np.float32)
    pd = np.clip(pts[:, None, :] + offs[None], -np.float32(radius), np.float32(radius))
    sd = net(pd.reshape(-1, 3).astype(np.float32))[:, 0].view(-1, 6)
    grad = 0.5 * (sd[:, 0::2] - sd[:, 1::2]) / float(e)
    lap = (sd[:, 0::2] + sd[:, 1::2] - 2 * sdf[:, None]).sum(-1) / float(e) ** 2
    return (sdf * d[0]).sum() + (grad * d[1]).sum() + (out * d[2]).sum() + (lap * d[3]).sum()


def test_sdf_fd_bwd(dev):
    tab = _table(11, 0.5)
    mlp = _mlp(12)
    n = 777   # not a multiple of 64: exercises the inactive-lane path
    pts = _pts(n, 13, -1.0, 1.0)
    eps, active, radius = 0.031, 5, 1.0
    g = torch.Generator().manual_seed(14)
    d = [torch.randn(n, generator=g), torch.randn(n, 3, generator=g) * 0.1,
         torch.randn(n, 13, generator=g), torch.randn(n, generator=g) * 1e-3]
    tab64 = tab.double().requires_grad_(True)
    mlp64 = [m.double().requires_grad_(True) for m in mlp]
    loss = _torch_fd_loss(tab64, mlp64, pts.numpy(), eps, active, radius, [x.double() for x in d])
    loss.backward()
    gt, gm = ops.sdf_fd_bwd(CFG, tab.to(dev), [m.to(dev) for m in mlp], pts.to(dev), radius, eps,
                            active, *[x.to(dev) for x in d])
    gt = gt.cpu().numpy().reshape(-1, 2)
    ref_t = tab64.grad.numpy()
    assert np.array_equal(ref_t != 0, gt != 0)          # same entries touched
    scale = np.abs(ref_t).max()
    np.testing.assert_allclose(gt, ref_t, rtol=1e-4, atol=1e-5 * scale)
    for got, ref in zip(gm, mlp64):
        r = ref.grad.numpy()
        np.testing.assert_allclose(got.cpu().numpy(), r, rtol=1e-4,
                                   atol=1e-5 * max(np.abs(r).max(), 1.0))


def test_sdf_fd_bwd_from_the_feature_cache_vs_float64_autograd(dev):
    """The path the optimisation runs — forward with the feature cache, backward from it (for 4..7
    active levels the lean / pipelined kernel sdf_fd_bwd_pipe_kernel) — against float64 autograd."""
    tab = _table(15, 0.5)
    mlp = _mlp(16)
    n = 1000 + 37
    pts = _pts(n, 17, -1.0, 1.0)
    eps, active, radius = 0.031, 5, 1.0
    g = torch.Generator().manual_seed(18)
    d = [torch.randn(n, generator=g), torch.randn(n, 3, generator=g) * 0.1,
         torch.randn(n, 13, generator=g), torch.randn(n, generator=g) * 1e-3]
    tab64 = tab.double().requires_grad_(True)
    mlp64 = [m.double().requires_grad_(True) for m in mlp]
    _torch_fd_loss(tab64, mlp64, pts.numpy(), eps, active, radius, [x.double() for x in d]).backward()
    tabd, mlpd, ptsd = tab.to(dev), [m.to(dev) for m in mlp], pts.to(dev)
    fwd = ops.sdf_fd_fwd(CFG, tabd, mlpd, ptsd, radius, eps, active, enc_cache=True)
    gt, gm = ops.sdf_fd_bwd(CFG, tabd, mlpd, ptsd, radius, eps, active, *[x.to(dev) for x in d],
                            enc_cache=fwd[4])
    gt = gt.cpu().numpy().reshape(-1, 2)
    ref_t = tab64.grad.numpy()
    assert np.array_equal(ref_t != 0, gt != 0)
    # the pipelined kernel recomputes layer 0 and forms dIn / the contractions over the points with three
    # bf16 products per f32 product (2^-16 each): measured against float64 (tools/sdf_bwd_accuracy.py)
    # rel-L2 9e-6 (table), 1.2e-5 / 1.8e-5 (W0, b0), 9e-6 (W1); largest element error 2.8e-5 of the
    # largest entry (the all-f32 MLP part: 5e-6 / 7e-6 / 5e-6 / 8e-6, 1.4e-5)
    np.testing.assert_allclose(gt, ref_t, rtol=1e-4, atol=4e-5 * np.abs(ref_t).max())
    assert np.linalg.norm(gt - ref_t) <= 3e-5 * np.linalg.norm(ref_t)
    for got, ref in zip(gm, mlp64):
        r = ref.grad.numpy()
        np.testing.assert_allclose(got.cpu().numpy(), r, rtol=1e-4,
                                   atol=5e-5 * max(np.abs(r).max(), 1.0))
        assert np.linalg.norm(got.cpu().numpy() - r) <= 5e-5 * np.linalg.norm(r)


@pytest.mark.parametrize("active", [4, 5, 6, 7])
@pytest.mark.parametrize("n", [1, 33, 256, 256 * 9, 64 * 256 + 5, 256 * 96 + 40, 256 * 256 * 2,
                               256 * 288 + 17, 70001])
def test_sdf_fd_bwd_pipelined_kernel_equals_the_general_kernel(dev, active, n):
    """sdf_fd_bwd_pipe_kernel (backward from the feature cache, 4..7 active levels) against the general
    kernel (exact f32 MLP part): the pipelined kernel forms layer 0 of its recompute (4..6 levels), dIn and
    the contractions over the points with three bf16 products per f32 product.  The general kernel is
    reached by the call WITHOUT a cache, which re-gathers the same f16 features; in that form it
    does not share the last partial iteration of a workgroup's range between the waves, so:
      * where every range is whole 256-point iterations (n = 256 k up to 256 workgroups, or a
        multiple of 256 * 256) g_b1 of the two is bit-identical (the weight gradients: bf16 x 3
        products in the pipelined kernel, 6e-5 of the largest entry);
      * otherwise (fewer points than workgroups, a last partial iteration of <= 64 / <= 128 points
        whose evaluations the waves share, a lone second half) the same per-point terms are summed
        by different waves: equal to float summation order (measured <= 1.1e-5 of the largest
        entry, tools/bwd_pipe_diff.py).
    The table gradients differ by the order of the float atomics of the scatter only."""
    tab = _table(61, 0.5).to(dev)
    mlp = [m.to(dev) for m in _mlp(62)]
    pts = _pts(n, 63 + n % 7, -1.0, 1.0).to(dev)
    eps, radius = 1.0 / 128, 1.0
    g = torch.Generator().manual_seed(64)
    d = [torch.randn(n, generator=g).to(dev), (torch.randn(n, 3, generator=g) * 0.1).to(dev),
         (torch.randn(n, 13, generator=g) * 0.1).to(dev), (torch.randn(n, generator=g) * 1e-4).to(dev)]
    fwd = ops.sdf_fd_fwd(CFG, tab, mlp, pts, radius, eps, active, enc_cache=True)
    gt0, gm0 = ops.sdf_fd_bwd(CFG, tab, mlp, pts, radius, eps, active, *d)
    gt1, gm1 = ops.sdf_fd_bwd(CFG, tab, mlp, pts, radius, eps, active, *d, enc_cache=fwd[4])
    whole = n % 256 == 0 and (n <= 256 * 256 or n % (256 * 256) == 0)
    for k, (a_, b_) in enumerate(zip(gm0, gm1)):
        # (g_w0, g_b0, g_w1: the contractions over the points run as bf16 x 3 in the pipelined
        # kernel — 2^-16 per product; g_b1 comes from identical operations)
        if whole and k == 3:
            assert torch.equal(a_, b_)
        else:
            # a bf16 x 3 product is good to ~2^-16 of ITSELF; with a handful of points an entry is a sum of
            # seven evaluations' terms of alternating sign (+-0.5 d_grad / eps), larger than the entry
            tol = 5e-4 if n < 1000 else 1e-4
            assert float((a_ - b_).abs().max()) <= tol * float(a_.abs().max()), (k, n, active)
    # (the same entries are touched; an entry whose float atomics cancel to exactly 0.0 in one order
    # and to a rounding residue in the other is covered by the bound on the difference)
    mism = (gt0 != 0) ^ (gt1 != 0)
    assert int(mism.sum()) <= 8
    assert float((gt0 - gt1).abs().max()) <= (5e-4 if n < 1000 else 1e-4) * float(gt0.abs().max())


def test_sdf_fd_bwd_points_outside_the_box(dev):
    """Points beyond [-radius, radius] (the perturbed random points of the sparsity / smoothness
    terms can be): fd_point clamps every coordinate of the six offset evaluations but not the
    centre (geometry.py:160-171), so the offsets do not share the centre's cell — the scatter's
    same-cell shortcut must not apply to them."""
    tab = _table(31, 0.5)
    mlp = _mlp(32)
    n = 500
    pts = _pts(n, 33, -1.25, 1.25)
    assert int((pts.abs() > 1).any(1).sum()) > 100
    eps, active, radius = 0.02, 4, 1.0
    g = torch.Generator().manual_seed(34)
    d = [torch.randn(n, generator=g), torch.randn(n, 3, generator=g) * 0.1,
         torch.randn(n, 13, generator=g), torch.randn(n, generator=g) * 1e-3]
    tab64 = tab.double().requires_grad_(True)
    mlp64 = [m.double().requires_grad_(True) for m in mlp]
    _torch_fd_loss(tab64, mlp64, pts.numpy(), eps, active, radius, [x.double() for x in d]).backward()
    gt, _ = ops.sdf_fd_bwd(CFG, tab.to(dev), [m.to(dev) for m in mlp], pts.to(dev), radius, eps,
                           active, *[x.to(dev) for x in d])
    gt = gt.cpu().numpy().reshape(-1, 2)
    ref_t = tab64.grad.numpy()
    np.testing.assert_allclose(gt, ref_t, rtol=1e-4, atol=1e-5 * np.abs(ref_t).max())


def test_sdf_fd_feature_cache_round_trip(dev):
    """Forward with the feature cache == plain forward (bit for bit); the cache holds exactly
    the forward's f16 features; backward from the cache == backward that re-gathers, up to the
    order of the float atomics (the MLP inputs are identical bits)."""
    tab = _table(21, 0.5).to(dev)
    mlp = [m.to(dev) for m in _mlp(22)]
    n = 5000 + 13
    pts = _pts(n, 23, -1.0, 1.0).to(dev)
    eps, active, radius = 0.027, 6, 1.0
    plain = ops.sdf_fd_fwd(CFG, tab, mlp, pts, radius, eps, active)
    cached = ops.sdf_fd_fwd(CFG, tab, mlp, pts, radius, eps, active, enc_cache=True)
    for a_, b_ in zip(plain, cached[:4]):
        assert torch.equal(a_, b_)
    cache = cached[4].view(7, n, active, 2)
    enc = ops.hashgrid_encode_fwd(CFG, tab, ((pts + radius) / (2 * radius)).clamp(0, 1), active)
    assert torch.equal(cache[0].reshape(n, -1), enc[:, :2 * active])
    g = torch.Generator().manual_seed(24)
    d = [torch.randn(n, generator=g).to(dev), (torch.randn(n, 3, generator=g) * 0.1).to(dev),
         torch.randn(n, 13, generator=g).to(dev), (torch.randn(n, generator=g) * 1e-3).to(dev)]
    gt0, gm0 = ops.sdf_fd_bwd(CFG, tab, mlp, pts, radius, eps, active, *d)
    gt1, gm1 = ops.sdf_fd_bwd(CFG, tab, mlp, pts, radius, eps, active, *d, enc_cache=cached[4])
    # (the backward from the cache is the pipelined kernel for 4..7 levels: three bf16 products per f32
    # product in its recompute of layer 0 and its point contractions — tools/sdf_bwd_accuracy.py)
    scale = float(gt0.abs().max())
    assert float((gt0 - gt1).abs().max()) < 5e-5 * scale
    assert int(((gt0 != 0) ^ (gt1 != 0)).sum()) <= 8
    for a_, b_ in zip(gm0, gm1):
        assert float((a_ - b_).abs().max()) < 1e-4 * (float(a_.abs().max()) + 1e-12)


@pytest.mark.parametrize("active", [4, 5, 6, 7])
def test_sdf_fd_fwd_seven_evaluations_bit_exact(dev, active):
    """The level-outer forward (corners of the centre's cell shared with the +-eps evaluations,
    csrc/hashgrid.hip) against the same seven evaluations done one by one with the plain kernels:
    the cached f16 features of every evaluation equal dsu_hashgrid_encode_fwd on the offset point
    (geometry.py:160-171: offsets, clamp to the box, contraction), bit for bit, and sdf / feature
    equal dsu_sdf_fwd's.  eps follows the reference's progressive rule (geometry.py:196-215:
    the finest active level's cell) so that the shared-corner path is the one that runs; the rows
    appended behind it take the fix-up path too: on the box faces, outside the box."""
    radius = 1.0
    tab = _table(91, 0.5).to(dev)
    mlp = [m.to(dev) for m in _mlp(92)]
    g = torch.Generator().manual_seed(93 + active)
    pts = torch.rand(20000 + 37, 3, generator=g) * 2 - 1
    edge = torch.tensor([[1.0, -1.0, 0.3], [-1.0, 1.0, 1.0], [0.99999, 0.0, -0.99999],
                         [1.2, 0.1, 0.1], [0.0, -1.5, 0.7], [3.0, 3.0, -3.0]])
    pts = torch.cat([pts, edge]).to(dev)
    n = pts.shape[0]
    eps = 2.0 * radius / (CFG.base_resolution * CFG.per_level_scale ** (active - 1))
    sdf, grad, feat, lap, cache = ops.sdf_fd_fwd(CFG, tab, mlp, pts, radius, eps, active,
                                                 enc_cache=True)
    cache = cache.view(7, n, active, 2)
    e32 = torch.tensor(eps, dtype=torch.float32, device=dev)
    s = []
    for e in range(7):
        q = pts.clone()
        if e > 0:
            ax = (e - 1) // 2
            q[:, ax] = q[:, ax] + (-e32 if (e - 1) & 1 else e32)
            q = q.clamp(-radius, radius)
        x = (q + radius) / (2 * radius)              # scale_anything(x, (-r, r), (0, 1))
        enc = ops.hashgrid_encode_fwd(CFG, tab, x, active)
        assert torch.equal(cache[e].reshape(n, -1), enc[:, :2 * active]), e
        out = ops.sdf_fwd(CFG, tab, mlp, q, radius, active, n_out=13)
        s.append(out[:, 0])
        if e == 0:
            assert torch.equal(feat, out)
    assert torch.equal(sdf, s[0])
    ref_grad = torch.stack([0.5 * (s[1] - s[2]) / e32, 0.5 * (s[3] - s[4]) / e32,
                            0.5 * (s[5] - s[6]) / e32], 1)
    assert torch.equal(grad, ref_grad)


def test_sdf_full_size_linearity(dev):
    """BASELINE size (2^21-point export chunk): property check instead of the slow oracle.
    The network is affine in the second-layer bias: out(b1 + c) - out(b1) == c."""
    tab = _table(15, 0.5).to(dev)
    mlp = [m.to(dev) for m in _mlp(16)]
    pts = _pts(2097152, 17, -1.0, 1.0).to(dev)
    a = ops.sdf_fwd(CFG, tab, mlp, pts, 1.0, 7, 1)
    mlp2 = [mlp[0], mlp[1], mlp[2], mlp[3] + 0.25]
    b = ops.sdf_fwd(CFG, tab, mlp2, pts, 1.0, 7, 1)
    assert torch.isfinite(a).all()
    torch.testing.assert_close(b - a, torch.full_like(a, 0.25), rtol=0, atol=1e-5)
    # and a sampled subset against the oracle
    sel = torch.arange(0, 2097152, 4099)
    ref = oh.sdf_network(tab.cpu().numpy(), [m.cpu().numpy() for m in mlp],
                         pts[sel].cpu().numpy(), 1.0, LV, 7)[:, :1]
    np.testing.assert_allclose(a[sel].cpu().numpy(), ref, rtol=2e-5, atol=2e-5)


def test_fused_sdf_matches_reference_volume_sdf_fixture(dev):
    """The fused kernel against outputs of the reference's own VolumeSDF.forward
    (tests/golden/nsr_reference.npz, tests/golden/make_nsr_golden.py)."""
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "nsr_reference.npz"))
    g = torch.Generator().manual_seed(int(gold["table_seed"]))
    tab = ((torch.rand(CFG.n_params, generator=g) * 2 - 1) * float(gold["table_scale"])).half()
    tab = tab.view(-1, 2).to(dev)
    mlp = [torch.from_numpy(gold[k]).to(dev) for k in ("w0", "b0", "w1", "b1")]
    for step, level in ((0, 4), (1500, 5), (2999, 6)):
        k = f"s{step}."
        eps = float(gold[k + "eps"])
        pts = torch.from_numpy(gold[k + "pts"]).to(dev)
        sdf, grad, feat, lap = ops.sdf_fd_fwd(CFG, tab, mlp, pts, 1.0, eps, level)
        np.testing.assert_allclose(sdf.cpu().numpy(), gold[k + "sdf"], rtol=0, atol=5e-6)
        np.testing.assert_allclose(feat.cpu().numpy(), gold[k + "feature"], rtol=0, atol=5e-6)
        np.testing.assert_allclose(grad.cpu().numpy(), gold[k + "grad"], rtol=0, atol=2e-6 / eps)
        np.testing.assert_allclose(lap.cpu().numpy(), gold[k + "laplace"], rtol=0, atol=2e-5 / eps ** 2)
        s1 = ops.sdf_fwd(CFG, tab, mlp, pts, 1.0, level, 1)[:, 0]
        np.testing.assert_allclose(s1.cpu().numpy(), gold[k + "forward_level"], rtol=0, atol=5e-6)


def test_sdf_fd_bwd_fused_form_in_subprocess():
    """The single-kernel form of the backward (DSU_BWD_SPLIT=0; the default is the two-kernel form)
    is selected once per process, so it is checked in a child process.  The switch exists only in
    variant builds of the library (-DDSU_AB_SWITCHES, loaded through DSU_HIP_LIB by tools/): with
    the product library this test has nothing to select."""
    import os, subprocess, sys
    from drawingspinup_amd import _lib
    if not _lib.lib().dsu_ab_switches():
        pytest.skip("product library: no A/B switches compiled in")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSU_BWD_SPLIT="0", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu",
                        os.path.join(root, "tests", "test_gpu_hashgrid.py") + "::test_sdf_fd_bwd",
                        os.path.join(root, "tests", "test_gpu_hashgrid.py")
                        + "::test_sdf_fd_feature_cache_round_trip"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_sdf_fd_bwd_rejects_resolutions_beyond_the_cell_key(dev):
    """The same-cell run merge packs cell coordinates into 10 bits each: a grid whose active levels
    exceed 1023 cells per axis must be refused, not silently aliased."""
    cfg = ops.HashGridConfig(n_levels=12, base_resolution=64)
    n = 64
    tab = torch.zeros(cfg.n_entries, 2, dtype=torch.float16, device=dev)
    mlp = [torch.zeros(64, 27), torch.zeros(64), torch.zeros(13, 64), torch.zeros(13)]
    mlp = [m.to(dev) for m in mlp]
    pts = _pts(n, 3, -1.0, 1.0).to(dev)
    d = [torch.zeros(n, device=dev), torch.zeros(n, 3, device=dev), torch.zeros(n, 13, device=dev),
         torch.zeros(n, device=dev)]
    ops.sdf_fd_bwd(cfg, tab, mlp, pts, 1.0, 0.01, 10, *d)          # level 9: 64 * s^9 = 776 cells
    with pytest.raises(Exception):
        ops.sdf_fd_bwd(cfg, tab, mlp, pts, 1.0, 0.01, 12, *d)      # level 11: 1353 cells


# ---------------------------------------------------------------- sorted evaluation order
def _morton_bins(pts, radius, bits):
    nb = 1 << bits
    c = np.clip(np.floor((pts.astype(np.float32) + np.float32(radius)) *
                         np.float32(1.0 / (2.0 * radius)) * np.float32(nb)), 0, nb - 1).astype(np.int64)
    key = np.zeros(len(pts), np.int64)
    for b in range(bits):
        for a in range(3):
            key |= ((c[:, a] >> b) & 1) << (3 * b + a)
    return key


@pytest.mark.parametrize("n,bits", [(1, 6), (777, 4), (70001, 6), (70001, 7)])
def test_spatial_sort_is_a_morton_ordered_permutation(dev, n, bits):
    pts = _pts(n, 31, -1.0, 1.0)
    pts[0] = torch.tensor([1.0, -1.0, 1.0])          # cube corners clamp into the last / first bin
    ps, perm = ops.spatial_sort(pts.to(dev), 1.0, bits)
    perm = perm.cpu().numpy()
    assert np.array_equal(np.sort(perm), np.arange(n))                 # INT: a permutation
    assert np.array_equal(ps.cpu().numpy(), pts.numpy()[perm])          # rows moved bit for bit
    key = _morton_bins(pts.numpy()[perm], 1.0, bits)
    assert (np.diff(key) >= 0).all()                                    # bins in Morton order


def test_spatial_sort_empty(dev):
    ps, perm = ops.spatial_sort(torch.empty(0, 3, device=dev), 1.0)
    assert ps.shape == (0, 3) and perm.shape == (0,)


@pytest.mark.parametrize("active", [4, 6])
def test_sorted_order_gives_the_same_forward_and_backward(dev, active):
    """dsu_sdf_fd_{fwd,bwd}_sorted == the plain calls: per-point outputs bit for bit (rows come
    back in the caller's order), gradients to summation-order rounding."""
    n = 40000
    tab = _table(41, 0.5).to(dev)
    mlp = [m.to(dev) for m in _mlp(42)]
    pts = (_pts(n, 43, -0.7, 0.7)).to(dev)
    g = torch.Generator().manual_seed(44)
    d_sdf = torch.randn(n, generator=g).to(dev)
    d_grad = (torch.randn(n, 3, generator=g) * 0.1).to(dev)
    d_feat = (torch.randn(n, 13, generator=g) * 0.1).to(dev)
    eps = 1.0 / 128
    ref = ops.sdf_fd_fwd(CFG, tab, mlp, pts, 1.0, eps, active, True, True, True, enc_cache=True)
    gt_ref, g_ref = ops.sdf_fd_bwd(CFG, tab, mlp, pts, 1.0, eps, active, d_sdf, d_grad, d_feat,
                                   None, enc_cache=ref[4])
    ps, perm = ops.spatial_sort(pts, 1.0, 6)
    got = ops.sdf_fd_fwd(CFG, tab, mlp, ps, 1.0, eps, active, True, True, True, enc_cache=True,
                         perm=perm)
    for a, b in zip(ref[:4], got[:4]):
        assert torch.equal(a, b)
    gt, gg = ops.sdf_fd_bwd(CFG, tab, mlp, ps, 1.0, eps, active, d_sdf, d_grad, d_feat, None,
                            enc_cache=got[4], perm=perm)
    assert torch.equal(gt_ref != 0, gt != 0)                            # same touched entries
    scale = float(gt_ref.abs().max())
    np.testing.assert_allclose(gt.cpu().numpy(), gt_ref.cpu().numpy(), rtol=1e-4,
                               atol=2e-6 * scale)
    for a, b in zip(g_ref, gg):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-4,
                                   atol=2e-5 * float(a.abs().max()))


@pytest.mark.parametrize("kind", ["ray", "repeat"])
def test_sdf_fd_bwd_on_samples_that_share_cells(dev, kind, monkeypatch):
    """Neighbouring lanes in the SAME cell (samples along a ray, repeated points): the same-cell
    run merge of the backward scatter against float64 autograd.  (Uniform random points never
    put two neighbouring lanes into one cell; a DPP move under a short-circuit's EXEC mask once
    dropped / doubled the row-end lanes of such runs without any test noticing.)"""
    tab = _table(51, 0.5)
    mlp = _mlp(52)
    if kind == "ray":
        t = torch.linspace(0.0, 0.45, 333)
        pts = torch.cat([torch.tensor([[0.12, -0.3, -0.2]]) + t[:, None] * torch.tensor([[0.1, 0.2, 1.0]]),
                         torch.tensor([[-0.4, 0.33, 0.5]]) - t[:, None] * torch.tensor([[0.0, 1.0, 0.3]])])
    else:
        pts = torch.cat([(torch.tensor([[0.1234, -0.3, 0.21]]) + 0.11 * k).repeat(r, 1)
                         for k, r in enumerate((1, 2, 15, 16, 17, 33, 64, 5))])
    n = pts.shape[0]
    eps, active, radius = 1.0 / 128, 5, 1.0
    g = torch.Generator().manual_seed(53)
    d = [torch.randn(n, generator=g), torch.randn(n, 3, generator=g) * 0.1,
         torch.randn(n, 13, generator=g) * 0.1, torch.randn(n, generator=g) * 1e-4]
    tab64 = tab.double().requires_grad_(True)
    mlp64 = [m.double().requires_grad_(True) for m in mlp]
    _torch_fd_loss(tab64, mlp64, pts.numpy(), eps, active, radius, [x.double() for x in d]).backward()
    ref_t = tab64.grad.numpy()
    scale = np.abs(ref_t).max()
    from drawingspinup_amd import _lib
    # the fused single-kernel form only exists behind the variant builds' DSU_BWD_SPLIT switch
    for split in (("1", "0") if _lib.lib().dsu_ab_switches() else ("1",)):
        monkeypatch.setenv("DSU_BWD_SPLIT", split)
        gt, gm = ops.sdf_fd_bwd(CFG, tab.to(dev), [m.to(dev) for m in mlp], pts.to(dev), radius, eps,
                                active, *[x.to(dev) for x in d])
        gt = gt.cpu().numpy().reshape(-1, 2)
        assert np.array_equal(ref_t != 0, gt != 0)
        np.testing.assert_allclose(gt, ref_t, rtol=1e-4, atol=1e-5 * scale)
        for got, ref in zip(gm, mlp64):
            r = ref.grad.numpy()
            np.testing.assert_allclose(got.cpu().numpy(), r, rtol=1e-4,
                                       atol=1e-5 * max(np.abs(r).max(), 1.0))
