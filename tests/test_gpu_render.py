"""HIP ray marching / compositing kernels vs the nerfacc restatement."""
import numpy as np
import pytest
import torch

from drawingspinup_amd import ops
from oracle import nerfacc_ref as nr

pytestmark = pytest.mark.gpu
AABB = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]
STEP = 1.732 * 2 * 1.0 / 1024


def _sphere_grid(res=128, r=0.5):
    c = (np.arange(res) + 0.5) / res * 2 - 1
    x, y, z = np.meshgrid(c, c, c, indexing="ij")
    return ((x * x + y * y + z * z) <= r * r).reshape(-1).astype(np.uint8)


def _ortho_rays(n, seed, axis=2, sign=1.0):
    g = np.random.default_rng(seed)
    uv = (g.random((n, 2)).astype(np.float32) - 0.5) * 2 * 0.9
    o = np.zeros((n, 3), np.float32)
    d = np.zeros((n, 3), np.float32)
    oth = [a for a in range(3) if a != axis]
    o[:, oth[0]] = uv[:, 0]; o[:, oth[1]] = uv[:, 1]; o[:, axis] = -1.3 * sign
    d[:, axis] = sign
    return o, d


@pytest.mark.parametrize("axis,sign", [(2, 1.0), (0, -1.0), (1, 1.0)])
def test_march_matches_oracle_bit_exact(dev, axis, sign):
    occ = _sphere_grid()
    o, d = _ortho_rays(300, 3 + axis, axis, sign)
    tmin_ref, tmax_ref = nr.ray_aabb_intersect(o, d, AABB)
    to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    tmin, tmax = ops.ray_aabb(to, td, AABB)
    assert np.array_equal(tmin.cpu().numpy(), tmin_ref) and np.array_equal(tmax.cpu().numpy(), tmax_ref)
    ri, ts, te, off, cnt = ops.ray_march(to, td, tmin, tmax, AABB, torch.from_numpy(occ).to(dev),
                                         128, STEP)
    ri_r, ts_r, te_r, cnt_r = nr.ray_marching(o, d, tmin_ref, tmax_ref, AABB, occ, 128, STEP)
    assert cnt_r.sum() > 1000
    assert np.array_equal(cnt.cpu().numpy(), cnt_r)            # packed counts: INT, bit-exact
    assert np.array_equal(ri.cpu().numpy(), ri_r)              # ray_indices: INT, bit-exact
    assert np.array_equal(ts.cpu().numpy(), ts_r) and np.array_equal(te.cpu().numpy(), te_r)


def test_march_oblique_and_jitter(dev):
    occ = _sphere_grid()
    g = np.random.default_rng(0)
    n = 200
    o = np.tile(np.array([[0.1, -0.2, -1.5]], np.float32), (n, 1))
    d = g.normal(size=(n, 3)).astype(np.float32) * 0.2 + np.array([0, 0, 1], np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    jit = g.random(n).astype(np.float32)
    tmin_ref, tmax_ref = nr.ray_aabb_intersect(o, d, AABB)
    tmin_j = (tmin_ref + jit * np.float32(STEP)).astype(np.float32)
    to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    tmin, tmax = ops.ray_aabb(to, td, AABB, torch.from_numpy(jit).to(dev), STEP)
    assert np.array_equal(tmin.cpu().numpy(), tmin_j)
    ri, ts, te, off, cnt = ops.ray_march(to, td, tmin, tmax, AABB, torch.from_numpy(occ).to(dev),
                                         128, STEP)
    ri_r, ts_r, te_r, cnt_r = nr.ray_marching(o, d, tmin_j, tmax_ref, AABB, occ, 128, STEP)
    assert np.array_equal(cnt.cpu().numpy(), cnt_r)
    assert np.array_equal(ts.cpu().numpy(), ts_r)


def test_march_empty_grid_and_no_grid(dev):
    o, d = _ortho_rays(64, 1)
    to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    tmin, tmax = ops.ray_aabb(to, td, AABB)
    empty = torch.zeros(128 ** 3, dtype=torch.uint8, device=dev)
    ri, ts, te, off, cnt = ops.ray_march(to, td, tmin, tmax, AABB, empty, 128, STEP)
    assert ri.numel() == 0 and int(cnt.sum()) == 0
    ri, ts, te, off, cnt = ops.ray_march(to, td, tmin, tmax, AABB, None, 0, STEP)
    # dense marching through the whole box: (far-near)/step samples per ray
    assert abs(int(cnt[0]) - int(2.0 / STEP)) <= 1
    # a ray that misses the box
    o2 = np.array([[3.0, 3.0, -1.3]], np.float32); d2 = np.array([[0, 0, 1.0]], np.float32)
    tmin2, tmax2 = ops.ray_aabb(torch.from_numpy(o2).to(dev), torch.from_numpy(d2).to(dev), AABB)
    assert float(tmin2[0]) == float(np.float32(1e10)) and float(tmax2[0]) == float(np.float32(1e10))


def test_weights_accumulate(dev):
    g = np.random.default_rng(5)
    counts = g.integers(0, 40, size=500).astype(np.int32)
    counts[7] = 0
    n = int(counts.sum())
    alpha = g.random(n).astype(np.float32)
    alpha[:5] = 1.0   # alpha == 1 kills transmittance: the backward clamp path
    ri = np.repeat(np.arange(500), counts)
    off = torch.from_numpy((np.cumsum(counts) - counts).astype(np.int32)).to(dev)
    cnt = torch.from_numpy(counts).to(dev)
    w = ops.weights_from_alpha_fwd(torch.from_numpy(alpha).to(dev), off, cnt)
    w_ref = nr.render_weight_from_alpha(alpha, counts)
    np.testing.assert_allclose(w.cpu().numpy(), w_ref, rtol=1e-5, atol=1e-7)
    gw = g.normal(size=n).astype(np.float32)
    ga = ops.weights_from_alpha_bwd(torch.from_numpy(alpha).to(dev), w, torch.from_numpy(gw).to(dev),
                                    off, cnt)
    ga_ref = nr.render_weight_from_alpha_bwd(alpha, counts, gw)
    ok = alpha < 0.99   # 1/(1-alpha) amplifies f32 rounding without bound as alpha -> 1
    np.testing.assert_allclose(ga.cpu().numpy()[ok], ga_ref[ok], rtol=2e-4, atol=5e-5)
    vals = g.normal(size=(n, 3)).astype(np.float32)
    acc = ops.accumulate_fwd(w, torch.from_numpy(vals).to(dev), off, cnt)
    np.testing.assert_allclose(acc.cpu().numpy(), nr.accumulate_along_rays(w_ref, vals, ri, 500),
                               rtol=1e-5, atol=1e-6)
    op = ops.accumulate_fwd(w, None, off, cnt)
    np.testing.assert_allclose(op.cpu().numpy(), nr.accumulate_along_rays(w_ref, None, ri, 500),
                               rtol=1e-5, atol=1e-6)
    # first sample alpha == 1 -> one-hot weights (hand KAT)
    a1 = torch.tensor([1.0, 0.3, 0.7], device=dev)
    w1 = ops.weights_from_alpha_fwd(a1, torch.zeros(1, dtype=torch.int32, device=dev),
                                    torch.full((1,), 3, dtype=torch.int32, device=dev))
    assert w1.tolist() == [1.0, 0.0, 0.0]


def test_occgrid(dev):
    g = np.random.default_rng(9)
    occs = g.random(4096).astype(np.float32) * 0.01
    occ = g.random(4096).astype(np.float32) * 0.02
    t = torch.from_numpy(occs).to(dev)
    ops.occgrid_ema(t, None, torch.from_numpy(occ).to(dev), 0.95)
    ref, refbin = nr.occgrid_update(occs, None, occ, 0.95, 0.001)
    assert np.array_equal(t.cpu().numpy(), ref)
    thre = min(float(t.mean()), 0.001)
    b = ops.occgrid_binarize(t, thre)
    assert np.array_equal(b.cpu().numpy().astype(bool), ref > np.float32(thre))
    idx = torch.from_numpy(g.permutation(4096)[:1000].astype(np.int64)).to(dev)
    occ2 = torch.from_numpy(g.random(1000).astype(np.float32)).to(dev)
    before = t.clone()
    ops.occgrid_ema(t, idx, occ2, 0.95)
    exp = before.clone()
    exp[idx] = torch.maximum(before[idx] * 0.95, occ2)
    assert torch.equal(t, exp)
    # cells listed several times: every old value is gathered before any is written (nerfacc's
    # `occs[idx] = max(occs[idx] * decay, occ)`), i.e. ONE decay, and the largest candidate stays
    dup = torch.from_numpy(g.integers(0, 64, 5000).astype(np.int64)).to(dev)
    occ3 = torch.from_numpy((g.random(5000) * 0.004).astype(np.float32)).to(dev)
    before = t.clone()
    ops.occgrid_ema(t, dup, occ3, 0.95)
    best = torch.full_like(before, -1.0).scatter_reduce(0, dup, occ3, "amax", include_self=True)
    exp = torch.where(best >= 0, torch.maximum(before * 0.95, best), before)
    assert torch.equal(t, exp)


def test_single_pass_march_equals_two_pass(dev):
    occ = torch.from_numpy(_sphere_grid()).to(dev)
    g = np.random.default_rng(11)
    o = np.tile(np.array([[0.05, -0.1, -1.6]], np.float32), (777, 1))
    d = g.normal(size=(777, 3)).astype(np.float32) * 0.25 + np.array([0, 0, 1], np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    tmin, tmax = ops.ray_aabb(to, td, AABB)
    a = ops.ray_march(to, td, tmin, tmax, AABB, occ, 128, STEP)
    b = ops.ray_march_single_pass(to, td, tmin, tmax, AABB, occ, 128, STEP)
    assert a[0].numel() > 1000
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # dense (no grid): longest chords, still within the scratch capacity
    a = ops.ray_march(to, td, tmin, tmax, AABB, None, 0, STEP)
    b = ops.ray_march_single_pass(to, td, tmin, tmax, AABB, None, 0, STEP)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("n", [1, 7, 1000, 8192, 8193, 20011])
def test_ray_offsets_scan(dev, n):
    """dsu_ray_offsets: exclusive scan of the per-ray sample counts + (total, max), INT exact."""
    from drawingspinup_amd._lib import check, lib, ptr, stream
    g = torch.Generator().manual_seed(n)
    counts = torch.randint(0, 600, (n,), generator=g, dtype=torch.int32)
    c = counts.to(dev)
    off = torch.empty_like(c)
    stats = torch.empty(2, dtype=torch.int32, device=dev)
    check(lib().dsu_ray_offsets(ptr(c), n, ptr(off), ptr(stats), stream()), "dsu_ray_offsets")
    ref = torch.cumsum(counts.long(), 0) - counts.long()
    assert torch.equal(off.cpu().long(), ref)
    assert stats.cpu().tolist() == [int(counts.sum()), int(counts.max())]
