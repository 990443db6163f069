"""The NSR optimisation step sequenced by the library (csrc/nsr_driver.hip, dsu_nsr_driver_step)
against the same step sequenced from Python (OrthoNeuSSystem.training_step_fused, itself pinned to
the reference's training_step by tests/test_gpu_nsr_reference_step.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

from drawingspinup_amd import _lib, ops
from drawingspinup_amd.nsr.system import OrthoData, OrthoNeuSSystem

pytestmark = pytest.mark.gpu


def _draws(dev, seed, step, n_rays, n_random, V=6, H=300, W=500):
    i64 = lambda: torch.empty(n_rays, dtype=torch.int64, device=dev)
    idx, x, y = i64(), i64(), i64()
    jit = torch.empty(n_rays, device=dev)
    pr, pe = torch.empty(n_random, 3, device=dev), torch.empty(n_random, 3, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    ops.check(_lib.lib().dsu_nsr_draws(seed, step, n_rays, V, H, W, p(idx), p(x), p(y), p(jit),
                                       n_random, p(pr), p(pe), ops.stream()), "dsu_nsr_draws")
    return idx, x, y, jit, pr, pe


def test_philox_draws_have_the_right_ranges_and_moments(dev):
    n = 1 << 16
    idx, x, y, jit, pr, pe = _draws(dev, 1234, 7, n, n)
    assert int(idx.min()) == 0 and int(idx.max()) == 5
    assert int(x.min()) == 0 and int(x.max()) == 499 and int(y.min()) == 0 and int(y.max()) == 299
    np.testing.assert_allclose(np.bincount(idx.cpu().numpy(), minlength=6) / n, 1 / 6, atol=0.01)
    assert 0.0 <= float(jit.min()) and float(jit.max()) < 1.0
    assert abs(float(jit.mean()) - 0.5) < 0.01 and abs(float(jit.var()) - 1 / 12) < 0.003
    assert -1.0 <= float(pr.min()) and float(pr.max()) < 1.0
    assert abs(float(pr.mean())) < 0.01 and abs(float(pr.var()) - 1 / 3) < 0.01
    assert abs(float(pe.mean())) < 0.01 and abs(float(pe.var()) - 1.0) < 0.02
    assert abs(float((pe ** 4).mean()) - 3.0) < 0.15                     # normal kurtosis
    c = np.corrcoef(pe.cpu().numpy().T)
    assert np.abs(c - np.eye(3)).max() < 0.02                              # independent components
    assert abs(float(torch.corrcoef(torch.stack([x.float(), y.float()]))[0, 1])) < 0.02
    # same (seed, step) -> same draws; another step or seed -> different ones
    again = _draws(dev, 1234, 7, n, n)
    assert all(torch.equal(a, b) for a, b in zip((idx, x, y, jit, pr, pe), again))
    assert not torch.equal(_draws(dev, 1234, 8, n, n)[3], jit)
    assert not torch.equal(_draws(dev, 1235, 7, n, n)[3], jit)
    # a prefix of the rays does not depend on how many are drawn
    assert torch.equal(_draws(dev, 1234, 7, 100, 10)[3], jit[:100])


def _system(dev, seed, mode):
    ds = OrthoData.synthetic_sphere(256, device=dev)
    sysm = OrthoNeuSSystem(device=dev, seed=seed)
    sysm.dataset = ds
    sysm.step_mode = mode
    return sysm, ds


def _inject(ds, n, seed, dev):
    g = torch.Generator().manual_seed(seed)
    return {"index": torch.randint(0, len(ds.all_masks), (n,), generator=g).to(dev),
            "x": torch.randint(0, ds.w, (n,), generator=g).to(dev),
            "y": torch.randint(0, ds.h, (n,), generator=g).to(dev),
            "jitter": torch.rand(n, generator=g).to(dev),
            "pts_random": (torch.rand(2048, 3, generator=g) * 2 - 1).to(dev),
            "perturb": torch.randn(2048, 3, generator=g).to(dev)}


def test_native_step_follows_the_python_sequenced_step(dev):
    """Same parameters, same injected draws, real learning rates: after every one of 6 steps the
    sample count, the ray count of the next step, the seven loss terms and (at the end) every
    parameter and the hash table agree.  Differences: summation order of atomics and of the
    weight-norm reductions only."""
    a, ds = _system(dev, 5, "fused")
    b, _ = _system(dev, 5, "native")
    b.model.load_state_dict(a.model.state_dict())
    for s in range(6):
        inj = _inject(ds, int(a.train_num_rays), 100 + s, dev)
        torch.manual_seed(1000 + s)          # the occupancy refresh of step 0 draws from torch's RNG
        ra = a.training_step_fused(dict(inj))
        torch.manual_seed(1000 + s)
        rb = b.training_step_native(dict(inj))
        assert ra["n_samples"] == rb["n_samples"] and ra["n_rays"] == rb["n_rays"]
        assert a.train_num_rays == b.train_num_rays
        for k in ("rgb_mse", "normal", "mask", "eikonal", "sparsity", "normal_smooth"):
            va, vb = float(ra[k]), float(rb[k])
            # (the ranking terms select rays by sorted error: one-ulp differences of the
            # weight-norm reductions can move a ray across the selection boundary)
            assert abs(va - vb) <= 1e-3 * max(abs(va), 1e-3), (s, k, va, vb)
        assert abs(float(ra["loss"]) - float(rb["loss"])) < 5e-4 * abs(float(ra["loss"]))
    assert b._native is not None and b._native.adam_step == 6
    pa, pb = dict(a.model.named_parameters()), dict(b.model.named_parameters())
    # Adam's update is scale-free (eps = 1e-15): an element whose true gradient is zero follows
    # rounding noise by +-lr per step on BOTH paths, so the comparison is statistical: mean
    # deviation and the share of elements that moved apart by more than a tenth of one Adam step
    for n in pa:
        x, y = pa[n].detach().float().reshape(-1), pb[n].detach().float().reshape(-1)
        lr = 1e-2 if n.startswith("texture") else 1e-3
        dev_ = (x - y).abs()
        assert float(dev_.mean()) < 0.02 * lr, (n, float(dev_.mean()))
        assert float((dev_ > 0.1 * lr).float().mean()) < 0.02, (n, float((dev_ > 0.1 * lr).float().mean()))
    eb = b.model.geometry.hashgrid
    assert torch.equal(eb.table_f16(), eb.params.detach().half())          # image follows the master


def test_native_training_converges_with_prefetch_and_refresh(dev):
    """The default path: own Philox draws, the next step's samples prefetched on the side stream,
    occupancy refresh every 16 steps (no prefetch across it), level switch not reached.  The loss
    falls as it does on the Python-sequenced path and the reconstructed volume is the sphere."""
    sysm, ds = _system(dev, 3, "native")
    losses, counts = [], []
    for s in range(150):
        r = sysm.training_step()
        counts.append((r["n_samples"], r["n_rays"]))
        if s % 10 == 4:
            losses.append(float(r["loss"]))
    assert sysm._native is not None and sysm._python_steps == 0
    assert np.isfinite(losses).all() and losses[-1] < 0.6 * losses[0]
    assert all(0 < n <= (1 << 19) for n, _ in counts)
    assert counts[-1][1] > counts[0][1]                       # dynamic ray count grew (neus_ortho.py:88-92)
    ref, _ = _system(dev, 3, "fused")
    for s in range(150):
        rr = ref.training_step_fused()
    assert abs(losses[-1] - float(rr["loss"])) < 0.35 * float(rr["loss"])
    coarse, fine, vmin, vmax = sysm.export_levels(128)
    vol = float((coarse <= 0).float().mean()) * 8.0
    assert 0.3 < vol < 0.75                                    # sphere r = 0.5: 0.524
    # two systems with the same seed draw the same rays: the native path is reproducible
    again, _ = _system(dev, 3, "native")
    c2 = [(lambda r: (r["n_samples"], r["n_rays"]))(again.training_step()) for _ in range(15)]
    assert c2 == counts[:15]


def test_native_step_timing_counters(dev):
    from drawingspinup_amd.nsr import system as S
    sysm, _ = _system(dev, 9, "native")
    S.native_timing["totals"].clear()
    S.native_timing["enabled"], S.native_timing["stride"] = True, 1
    try:
        for _ in range(5):
            sysm.training_step()
        sysm._native.flush_timing()
    finally:
        S.native_timing["enabled"] = False
    t = S.native_timing["totals"]
    print("native timing totals", t)
    assert t["sdf_fd_bwd"][0] == 5 and t["sdf_fd_fwd"][0] == 5      # stride 1: every step
    assert 0.05 < t["sdf_fd_bwd"][1] / 5 < 5.0                 # ms per launch
    assert t["sdf_fd_bwd"][2] > 5 * 50000 * (7 * 4 * 32 + 84) * 0.5      # algorithmic bytes
    # algorithmic MLP flops: backward = 3 x forward, forward = points x 2 x (7 x 64 x 11 + 64 x 19) at 4 levels
    assert abs(t["sdf_fd_bwd"][3] / t["sdf_fd_fwd"][3] - 3.0) < 1e-9
    pts = t["sdf_fd_fwd"][2] / (7 * 4 * 32 + 84)
    assert abs(t["sdf_fd_fwd"][3] / (pts * 2 * (7 * 64 * 11 + 64 * 19)) - 1.0) < 1e-6
    S.native_timing["totals"].clear()
    # sampled timing (what bench.py uses): only the steps whose index is a multiple of the stride
    S.native_timing["enabled"], S.native_timing["stride"] = True, 3
    try:
        first = int(sysm.global_step)
        for _ in range(7):
            sysm.training_step()
        sysm._native.flush_timing()
    finally:
        S.native_timing["enabled"], S.native_timing["stride"] = False, 1
    want = sum(1 for k in range(first, first + 7) if k % 3 == 0)
    assert S.native_timing["totals"]["sdf_fd_fwd"][0] == want
    S.native_timing["totals"].clear()


def test_native_step_results_are_snapshots_and_survive_a_step_jump(dev):
    """(a) What training_step returns stays what it was: the driver's two sets of loss terms are
    reused two steps later and partly zeroed one step later, the result is a copy.  (b) When the
    caller moves global_step (resume, tests) the sample-loss accumulators of the new step were not
    pre-zeroed by its predecessor: the driver zeroes them itself — eikonal / sparsity / smoothness
    equal the Python-sequenced step's, not twice them."""
    a, ds = _system(dev, 9, "fused")
    b, _ = _system(dev, 9, "native")
    b.model.load_state_dict(a.model.state_dict())
    kept = []
    for s in range(4):
        inj = _inject(ds, int(a.train_num_rays), 300 + s, dev)
        torch.manual_seed(2000 + s)
        a.training_step_fused(dict(inj))
        torch.manual_seed(2000 + s)
        r = b.training_step_native(dict(inj))
        kept.append((r, {k: float(r[k]) for k in ("rgb_mse", "eikonal", "sparsity", "loss")}))
    for r, snap in kept:                                   # read AFTER the later steps ran
        for k, v in snap.items():
            assert float(r[k]) == v, k
    # jump by an even number of steps: same parity of the term set, no pre-zeroing happened
    a.global_step += 4
    b.global_step += 4
    inj = _inject(ds, int(a.train_num_rays), 777, dev)
    torch.manual_seed(3000)
    ra = a.training_step_fused(dict(inj))
    torch.manual_seed(3000)
    rb = b.training_step_native(dict(inj))
    for k in ("eikonal", "sparsity", "normal_smooth", "rgb_mse"):
        va, vb = float(ra[k]), float(rb[k])
        assert abs(va - vb) <= 2e-3 * max(abs(va), 1e-3), (k, va, vb)


def test_native_occupancy_refresh_matches_the_torch_update(dev):
    """dsu_nsr_driver_occ_refresh (cells -> points -> SDF -> alpha -> EMA -> mean -> binary on the
    stream, no host round trip) against OccupancyGrid._update with the SAME cells and uniforms
    (nerfacc's rule restated in nsr/render.py, pinned to oracle/nerfacc_ref by test_gpu_render.py)."""
    sysm, ds = _system(dev, 11, "native")
    for _ in range(3):
        sysm.training_step()
    drv, m = sysm._native, sysm.model
    grid = m.occupancy_grid
    assert grid.native_refresh is not None and drv.stepped
    n = grid.num_cells
    g = torch.Generator().manual_seed(5)
    cells = torch.randperm(n, generator=g)[: n // 3].to(dev)          # distinct cells: no EMA races
    rand = torch.rand(cells.numel(), 3, generator=g).to(dev)
    occs0, bin0 = grid.occs.clone(), grid.binary_u8().clone()
    # torch path (injected cells switch the native hook off)
    grid._update(512, m.occ_eval_fn, occ_thre=0.01, ema_decay=0.95, rand=rand, indices=cells)
    occs_py, bin_py = grid.occs.clone(), grid.binary_u8().clone()
    thre_py = float(torch.clamp(occs_py.mean(), max=0.01))
    # native path from the same state
    grid.occs.copy_(occs0)
    grid._binary_u8 = bin0.clone()
    assert drv.occ_refresh(grid, 512, False, 0.01, 0.95, inj_cells=cells, inj_rand=rand)
    occs_nat, bin_nat = grid.occs, grid.binary_u8()
    # (the SDF differs in the last bits: torch's weight_norm vs the driver's effective weights)
    torch.testing.assert_close(occs_nat, occs_py, rtol=1e-5, atol=1e-6)
    assert float((occs_nat != occs0).float().mean()) > 0.2                # it did update
    border = (occs_py - thre_py).abs() < 3e-6
    assert torch.equal(bin_nat[~border], bin_py[~border])
    assert grid.binary.dtype == torch.bool and torch.equal(grid.binary.reshape(-1), bin_nat.bool())


def test_native_occupancy_refresh_own_draws(dev):
    """The library's own selection, replayed: the call exports the cells and uniforms it drew; the
    torch form of OccupancyGrid._update (nerfacc's rule restated in nsr/render.py, pinned to
    oracle/nerfacc_ref by test_gpu_render.py) run on exactly those from the same state gives the
    same grid.  Structure of the selection (neus.py:85-88 -> grid.py _update): warm-up = every cell
    once in order; regular = N/4 uniform draws, then the occupied cells in ascending order
    (torch.nonzero) — every one of them, since there are fewer than N/4 on this scene."""
    sysm, ds = _system(dev, 12, "native")
    for _ in range(2):
        sysm.training_step()
    drv, m = sysm._native, sysm.model
    grid = m.occupancy_grid
    n = grid.num_cells
    for all_cells in (True, False):
        occs0, bin0 = grid.occs.clone(), grid.binary_u8().clone()
        ex = {}
        assert drv.occ_refresh(grid, 640 + int(all_cells), all_cells, 0.01, 0.95, export=ex)
        occs_nat, bin_nat = grid.occs.clone(), grid.binary_u8().clone()
        cells, rand = ex["cells"].long(), ex["rand"]
        assert 0.0 <= float(rand.min()) and float(rand.max()) < 1.0
        if all_cells:
            assert torch.equal(cells, torch.arange(n, device=dev))
        else:
            uni, occ = cells[: n // 4], cells[n // 4:]
            assert int(uni.min()) >= 0 and int(uni.max()) < n
            assert abs(float(uni.float().mean()) / n - 0.5) < 0.01        # uniform over the grid
            assert float(torch.unique(uni).numel()) / uni.numel() > 0.85  # 4 (1 - e^-1/4) = 0.885
            was_on = torch.nonzero(bin0.bool())[:, 0]
            assert 0 < was_on.numel() <= n // 4
            assert torch.equal(occ[: was_on.numel()], was_on)            # all of them, in order
            assert bool((occ[was_on.numel():] == -1).all())              # the rest of the slots unused
        # replay through the torch form from the same state (explicit cells switch the hook off)
        valid = cells >= 0
        grid.occs.copy_(occs0)
        grid._binary_u8 = bin0.clone()
        grid._update(640 + int(all_cells), m.occ_eval_fn, occ_thre=0.01, ema_decay=0.95,
                     rand=rand[valid], indices=cells[valid])
        occs_py, bin_py = grid.occs.clone(), grid.binary_u8().clone()
        # (the SDF differs in the last bits: torch's weight_norm vs the driver's effective weights)
        torch.testing.assert_close(occs_nat, occs_py, rtol=1e-5, atol=1e-6)
        thre_py = float(torch.clamp(occs_py.mean(), max=0.01))
        border = (occs_py - thre_py).abs() < 3e-6
        assert torch.equal(bin_nat[~border], bin_py[~border])
        assert float((occs_nat != occs0).float().mean()) > 0.2            # it did update
        # continue from the library's result
        grid.occs.copy_(occs_nat)
        grid._binary_u8 = bin_nat
        grid._binary = bin_nat.view(grid.res, grid.res, grid.res).bool()
    # the same call twice from the same state: same draws, same grid (no run-order dependence)
    occs0, bin0 = grid.occs.clone(), grid.binary_u8().clone()
    assert drv.occ_refresh(grid, 700, False, 0.01, 0.95)
    first = grid.occs.clone()
    grid.occs.copy_(occs0)
    grid.binary_u8().copy_(bin0)
    assert drv.occ_refresh(grid, 700, False, 0.01, 0.95)
    assert torch.equal(grid.occs, first)
    # the step after a refresh runs (the driver re-marches with the new grid)
    sysm.training_step()


def test_native_occupancy_refresh_subsamples_a_dense_grid(dev):
    """More occupied cells than N/4 (grid.py _update: `if n < len(occupied_indices)`: N/4 of them
    drawn WITH replacement) — the branch a sphere scene never reaches.  The grid is made dense by
    hand (a ball of radius 0.85: 32 % of the cells); the exported selection is checked for its
    structure and replayed through the torch form from the same state."""
    sysm, ds = _system(dev, 13, "native")
    for _ in range(2):
        sysm.training_step()
    drv, m = sysm._native, sysm.model
    grid = m.occupancy_grid
    n, res = grid.num_cells, grid.res
    c = (torch.arange(res, device=dev) + 0.5) / res * 2 - 1
    ball = (c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2).sqrt() <= 0.85
    grid._binary = ball
    grid._binary_u8 = ball.reshape(-1).to(torch.uint8).contiguous()
    grid.occs.copy_(ball.reshape(-1).float() * 0.5 + 1e-4)
    was_on = torch.nonzero(ball.reshape(-1))[:, 0]
    assert was_on.numel() > n // 4
    occs0, bin0 = grid.occs.clone(), grid.binary_u8().clone()
    ex = {}
    assert drv.occ_refresh(grid, 800, False, 0.01, 0.95, export=ex)
    occs_nat, bin_nat = grid.occs.clone(), grid.binary_u8().clone()
    cells, rand = ex["cells"].long(), ex["rand"]
    assert cells.numel() == 2 * (n // 4)
    occ = cells[n // 4:]
    assert int(occ.min()) >= 0                                            # every slot used
    assert bool(ball.reshape(-1)[occ].all())                              # drawn from the occupied cells
    uniq = torch.unique(occ).numel()
    # n/4 draws with replacement from K cells hit K (1 - exp(-n / 4K)) distinct ones
    K = was_on.numel()
    expect = K * (1.0 - np.exp(-(n // 4) / K))
    assert abs(uniq - expect) < 0.02 * expect and uniq < occ.numel()
    assert not bool((occ[1:] >= occ[:-1]).all())                          # not the ordered list
    assert abs(float(occ.float().mean()) - float(was_on.float().mean())) < 0.01 * n   # uniform over them
    grid.occs.copy_(occs0)
    grid._binary_u8 = bin0.clone()
    grid._binary = bin0.view(res, res, res).bool()
    grid._update(800, m.occ_eval_fn, occ_thre=0.01, ema_decay=0.95, rand=rand, indices=cells)
    occs_py, bin_py = grid.occs.clone(), grid.binary_u8().clone()
    torch.testing.assert_close(occs_nat, occs_py, rtol=1e-5, atol=1e-6)
    thre_py = float(torch.clamp(occs_py.mean(), max=0.01))
    border = (occs_py - thre_py).abs() < 3e-6
    assert torch.equal(bin_nat[~border], bin_py[~border])
    # same call, same state: same draws (the subsample is keyed by (seed, step) too)
    grid.occs.copy_(occs0)
    grid._binary_u8 = bin0.clone()
    grid._binary = bin0.view(res, res, res).bool()
    ex2 = {}
    assert drv.occ_refresh(grid, 800, False, 0.01, 0.95, export=ex2)
    assert torch.equal(ex2["cells"], ex["cells"]) and torch.equal(grid.occs, occs_nat)


# ------------------------------------------------------------------------------------------------
# The library-sequenced step (what bench.py times) against the REFERENCE's own training_step and
# loss.backward() directly: tests/golden/nsr_step_reference.npz / nsr_grad_reference.npz, made by the
# imported reference modules (tests/golden/make_nsr_{step,grad}_golden.py).  The fixture's ray batch
# goes in through dsu_nsr_step_args.inj_rays..., its random points through inj_pts_random /
# inj_perturb; with zero learning rates the parameters stay put and the driver's first AdamW
# moments are (1 - beta1) x the step's gradients (dsu_nsr_driver_adam_moments).
# ------------------------------------------------------------------------------------------------
def test_native_step_matches_reference_training_step_and_backward_fixture(dev):
    import test_gpu_nsr_reference_step as R
    GOLD, GRAD, L = R.GOLD, R.GRAD, R.L
    sysm = OrthoNeuSSystem(device=dev, seed=0)
    ref_model = R._model(dev)
    sysm.model.load_state_dict(ref_model.state_dict())
    m = sysm.model
    m.occupancy_grid._binary = ref_model.occupancy_grid._binary
    m.occupancy_grid._binary_u8 = None
    m.occupancy_grid.every_n_step = lambda *a, **k: None      # the fixture's grid stays as given
    m.geometry.hashgrid.invalidate()
    m.randomized = False
    m.config["randomized"] = False
    sysm.dataset = OrthoData.synthetic_sphere(256, device=dev)  # resident tensors the driver binds; unread
    sysm.step_mode = "native"
    sysm.global_step = int(GOLD["step"])
    sysm.train_num_rays = int(GOLD["rays"].shape[0])
    sysm._base_lrs = [0.0 for _ in sysm._base_lrs]              # the step leaves the parameters alone
    sysm.keep_table_grad = True
    t = lambda k: torch.from_numpy(GOLD[k]).to(dev)
    batch = {k: t("batch." + k) for k in ("rays", "rgb", "normal", "mask", "cosines", "view_weights")}
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    last = sysm.training_step_native({"batch": batch, "pts_random": t("pts_random"),
                                      "perturb": t("perturb")})
    # sample count and the loss terms of the reference's training_step
    assert last["n_samples"] == int(GOLD["fwd.num_samples"][0])
    want = {"rgb_mse": float(GOLD["loss.loss_rgb_mse"]) * L.lambda_rgb_mse,
            "normal": float(GOLD["loss.loss_normal"]) * L.lambda_normal,
            "mask": float(GOLD["loss.loss_mask"]) * L.lambda_mask,
            "eikonal": float(GOLD["loss.loss_eikonal"]) * L.lambda_eikonal,
            "sparsity": float(GOLD["loss.loss_sparsity"]) * L.lambda_sparsity,
            "normal_smooth": float(GOLD["loss.loss_3d_normal_smooth"]) * L.lambda_3d_normal_smooth}
    for k, w in want.items():
        assert abs(float(last[k]) - w) <= 1e-4 * max(abs(w), 1e-6), (k, float(last[k]), w)
    want_loss = float(GRAD["loss.total"])
    assert abs(float(last["loss"]) - want_loss) < 2e-5 * abs(want_loss)
    for n, p in m.named_parameters():                          # lr = 0: nothing moved
        assert torch.equal(p.detach(), before[n]), n
    # gradients of the 13 small tensors from the first moments
    drv = sysm._native
    beta1 = float(sysm.config.optimizer.betas[0])
    base = int(_lib.lib().dsu_nsr_driver_adam_moments(drv.handle, 0)) - drv.workspace.data_ptr()
    n_small = sum(p.numel() for p in drv.params)
    mom = drv.workspace[base:base + 4 * n_small].view(torch.float32).clone() / (1.0 - beta1)
    names = {id(p): n for n, p in m.named_parameters()}
    rel = lambda got, w: float(np.linalg.norm((got - w).ravel()) / (np.linalg.norm(w.ravel()) + 1e-30))
    off, seen = 0, 0
    for p in drv.params:
        name = names[id(p)]
        got = mom[off:off + p.numel()].cpu().numpy().reshape(tuple(p.shape))
        off += p.numel()
        w = GRAD["grad." + name].reshape(got.shape)
        assert rel(got, w) < 1e-3, (name, rel(got, w))
        np.testing.assert_allclose(got, w, rtol=0, atol=2e-3 * np.abs(w).max())
        seen += 1
    assert seen == 13 and off == n_small
    # the hash table's gradient (left in place: table_p = NULL skips the fused table update)
    (tk,) = [k[len("gradnz_idx."):] for k in GRAD.files if k.startswith("gradnz_idx.")]
    got = m.geometry.hashgrid.params.grad.detach().cpu().numpy().reshape(-1).astype(np.float64)
    assert got.size == int(GRAD["gradnz_numel." + tk])
    w = np.zeros_like(got)
    w[GRAD["gradnz_idx." + tk]] = GRAD["gradnz_val." + tk]
    assert rel(got, w) < 1e-2, rel(got, w)
    big = float(np.abs(w).max())
    assert abs(np.count_nonzero(got) - np.count_nonzero(w)) < 1e-3 * np.count_nonzero(w)
    assert float(np.abs(got[w == 0]).max()) < 1e-5 * big
