"""Training / export loop of the NSR stage (recon.py + systems/neus_ortho.py).

Mirrors OrthoNeuSSystem (2_charactor_reconstructor/instant_nsr/systems/neus_ortho.py:13-200)
without PyTorch-Lightning: same per-step order — preprocess_data (random view / pixel ray
batch), update_step (level schedule, eps, occupancy refresh), forward, the 7 loss terms,
dynamic ray count, AdamW with the three parameter groups and the Constant->Exponential LR
schedule of configs/neuralangelo-ortho-wmask.yaml:101-127.
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops
from .model import Cfg, NeuSModel

DEFAULT_SYSTEM_CONFIG = Cfg({     # configs/neuralangelo-ortho-wmask.yaml:86-141
    "loss": {"lambda_rgb_mse": 0.5, "lambda_rgb_l1": 0.0, "lambda_mask": 1.0,
             "lambda_eikonal": 0.2, "lambda_normal": 1.0, "lambda_3d_normal_smooth": 1.0,
             "lambda_sparsity": 0.5, "sparsity_scale": 100.0, "geo_aware": True,
             "rgb_p_ratio": 0.8, "normal_p_ratio": 0.8, "mask_p_ratio": 0.9},
    "optimizer": {"lr": 0.01, "betas": (0.9, 0.99), "eps": 1e-15,
                  "params": {"geometry": 0.001, "texture": 0.01, "variance": 0.001}},
    "constant_steps": 500, "max_steps": 3000,
})

VIEWS = ["front", "front_right", "right", "back", "left", "front_left"]
# Morton bins per axis (2^bits) for the sorted evaluation order of the geometry network in the
# fused step (0 would keep the marcher's ray-major order).
_SPATIAL_SORT_BITS = 6

# capacity (samples) of the fixed-size buffers the prefetch path packs into: 2x the schedule's
# target of 2^18 samples per step; a step that exceeds it is re-packed on the main stream
_PACK_CAPACITY = 1 << 19

_AZIMUTH = {"front": 0.0, "front_right": 45.0, "right": 90.0, "back": 180.0, "left": 270.0,
            "front_left": 315.0}


def ideal_w2c(view):
    """Axis-aligned orthographic world->camera pose with the structure of
    instant_nsr/datasets/fixed_poses/000_<view>_RT.txt (synthetic stand-in: exact cos/sin, no
    f32 residue; diagonal views sit at distance 1.3*sqrt(2) as in those files)."""
    a = math.radians(_AZIMUTH[view])
    c, s = round(math.cos(a), 12), round(math.sin(a), 12)
    dist = 1.3 * (math.sqrt(2.0) if view.startswith(("front_", "back_")) else 1.0)
    return np.array([[c, s, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [s, -c, 0.0, -dist]], np.float64)


def rt_opengl2opencv(RT):                      # datasets/ortho.py:31-38
    R_bcam2cv = np.asarray([[1, 0, 0], [0, -1, 0], [0, 0, -1]], np.float32)
    return np.concatenate([R_bcam2cv @ RT[:3, :3], (R_bcam2cv @ RT[:3, 3])[:, None]], 1)


def inv_rt(RT):                                # datasets/ortho.py:48-51
    return np.linalg.inv(np.concatenate([RT, np.array([[0, 0, 0, 1]])], 0))[:3, :]


def ortho_rays_hw(W, H):                       # models/ray_utils.py:20-33
    i, j = np.meshgrid(np.arange(W, dtype=np.float32) + 0.5, np.arange(H, dtype=np.float32) + 0.5,
                       indexing="xy")
    i, j = torch.from_numpy(i), torch.from_numpy(j)
    origins = torch.stack([(i / W - 0.5) * 2, (j / H - 0.5) * 2, torch.zeros_like(i)], -1)
    directions = torch.stack([torch.zeros_like(i), torch.zeros_like(j), torch.ones_like(i)], -1)
    return origins, directions


def get_ortho_rays(origins, directions, c2w):  # models/ray_utils.py:36-58, (N,3) case
    rays_d = torch.matmul(c2w[:, :3, :3], directions[:, :, None]).squeeze(-1)
    rays_o = torch.matmul(c2w[:, :3, :3], origins[:, :, None]).squeeze(-1)
    return c2w[:, :3, 3].expand(rays_d.shape) + rays_o, rays_d


class OrthoData:
    """The tensors OrthoDatasetBase.setup keeps resident on the GPU (datasets/ortho.py:99-151)."""

    def __init__(self, images, masks, normals_world, c2w, device):
        # resident, contiguous (V,H,W,*) f32 arrays: what the fused ray-batch kernel gathers from
        self.all_images = images.float().to(device).contiguous()           # (V,H,W,3) in [0,1]
        self.all_masks = masks.float().to(device).contiguous()             # (V,H,W)
        self.all_normals_world = normals_world.float().to(device).contiguous()
        self.all_c2w = c2w.float().to(device).contiguous()                 # (V,3,4)
        V, H, W = self.all_masks.shape
        self.h, self.w = H, W
        o, d = ortho_rays_hw(W, H)
        self.origins = o[None].expand(V, -1, -1, -1).contiguous().to(device)
        self.directions = d[None].expand(V, -1, -1, -1).contiguous().to(device)
        self.view_weights = torch.ones(V, H, W, device=device)
        self.has_mask = True
        self.front_mask = None

    @staticmethod
    def synthetic_sphere(size=1024, radius=0.5, device="cuda", views=VIEWS):
        """6 orthographic views of a sphere; analytic normals in the
        front camera's frame, disc masks, smooth colour."""
        H = W = size
        o, _ = ortho_rays_hw(W, H)
        x, y = o[..., 0], o[..., 1]
        r2 = x * x + y * y
        mask = r2 <= radius * radius
        z = -torch.sqrt(torch.clamp(radius * radius - r2, min=0.0))   # towards the camera
        n_cam = torch.stack([x, y, z], -1) / radius                    # OpenCV camera frame
        n_cam = n_cam * mask[..., None]
        front_c2w = inv_rt(rt_opengl2opencv(ideal_w2c("front")))
        imgs, masks, normals, poses = [], [], [], []
        for v in views:
            c2w = inv_rt(rt_opengl2opencv(ideal_w2c(v)))
            R = torch.from_numpy(c2w[:3, :3]).float()
            n_world_true = n_cam @ R.T
            # stored normals are expressed in the FRONT view's system (load_a_prediction,
            # normal_system='front'): what mv diffusion predicts for this view, rotated by the
            # front c2w -> equals the true world normal for a consistent prediction
            normals.append(n_world_true)
            col = 0.5 + 0.4 * torch.stack([n_world_true[..., 0], n_world_true[..., 1],
                                           n_world_true[..., 2]], -1)
            imgs.append(col * mask[..., None] + (~mask[..., None]) * 1.0)
            masks.append(mask)
            poses.append(torch.from_numpy(c2w).float())
        del front_c2w
        return OrthoData(torch.stack(imgs), torch.stack(masks), torch.stack(normals),
                         torch.stack(poses), device)


def binary_cross_entropy(inp, target):         # systems/criterions.py:4-13 (reduction='none')
    return -(target * torch.log(inp) + (1 - target) * torch.log(1 - inp))


def ranking_loss(error, penalize_ratio=0.7, extra_weights=None, type="mean"):
    error, indices = torch.sort(error)         # systems/criterions.py:16-27
    k = int(penalize_ratio * indices.shape[0])
    s_error = torch.index_select(error, 0, index=indices[:k])
    if extra_weights is not None:
        s_error = s_error * torch.index_select(extra_weights, 0, index=indices[:k])
    return torch.mean(s_error) if type == "mean" else torch.sum(s_error)


def ranking_loss_masked(error, valid, penalize_ratio=0.7, extra_weights=None, type="mean"):
    """ranking_loss(error[valid], ratio, extra_weights[valid], type) without the boolean-mask
    gather: `x[mask]` makes the host wait for the device (nonzero), several times per step.

    NB the reference indexes the SORTED errors with the ORIGINAL positions of the k smallest
    (criterions.py:17-18: `error, indices = torch.sort(error)` then
    `index_select(error, 0, indices[:k])`), i.e. it sums sorted[p] over the subset positions p of
    the k smallest entries.  That exact selection is reproduced: masked-out entries sort to the
    end as +inf, subset positions come from a running count of `valid`, and
    k = int(ratio * valid.sum()) stays on the device (float64 product = Python's value).

    Ties: with that selection rule the result depends on the ORDER of equal errors whenever a run
    of ties straddles k (rays that miss the object all have opacity 0 and therefore equal mask
    errors), and the reference's `torch.sort` (stable=False) leaves that order to the backend.
    Here, as in the HIP kernel (64-bit (value, index) keys), ties keep their original order —
    torch.sort(stable=True); the reference-generated fixture is produced under the same rule."""
    n = error.shape[0]
    e = torch.where(valid, error, torch.full_like(error, float("inf")))
    se, idx = torch.sort(e, stable=True)                      # sorted valid errors, then +inf
    k = torch.floor(penalize_ratio * valid.sum().double())
    sel = torch.arange(n, device=error.device) < k
    pos = torch.cumsum(valid.to(torch.int64), 0) - 1          # position inside error[valid]
    pj = torch.index_select(pos, 0, idx).clamp_(min=0)        # subset position of the j-th smallest
    vals = torch.index_select(se, 0, pj)
    vals = torch.where(sel, vals, torch.zeros_like(vals))
    if extra_weights is not None:
        vals = vals * torch.index_select(extra_weights, 0, idx)
    total = vals.sum()
    return total / k.to(total.dtype) if type == "mean" else total


class _LazyLoss:
    """Total loss of a fused step, summed only when somebody looks at it (three launches that the
    optimisation itself never needs)."""

    def __init__(self, rterms, sterms):
        self._parts, self._value = (rterms, sterms), None

    def detach(self):
        if self._value is None:
            self._value = self._parts[0].sum() + self._parts[1].sum()
        return self._value

    def __float__(self):
        return float(self.detach())

    item = __float__


class RayLossGraph:
    """HIP-graph capture of OrthoNeuSSystem.ray_losses + d(loss)/d(comp) on fixed-capacity
    buffers.  The ray count changes every step (dynamic ray sampling), so the tensors are padded
    to `capacity` rows and the real rows are marked from a device-side count; replaying the graph
    is ONE host call instead of ~200 small launches."""

    KEYS = ("rgb", "normal", "mask", "cosines", "view_weights")

    def __init__(self, system, capacity):
        self.system, self.capacity = system, capacity
        dev = system.device
        z = lambda *shape: torch.zeros(*shape, device=dev)
        self.comp = z(capacity, 8).requires_grad_(True)
        self.bufs = {"rgb": z(capacity, 3), "normal": z(capacity, 3), "mask": z(capacity),
                     "cosines": z(capacity), "view_weights": z(capacity)}
        self.n_rays = torch.zeros(1, dtype=torch.int64, device=dev)
        self.graph = None
        self.names = None

    def _compute(self):
        pad = torch.arange(self.capacity, device=self.comp.device) < self.n_rays
        terms = self.system.ray_losses(self.comp, self.bufs, pad)
        self.names = list(terms.keys())
        total = sum(terms.values())
        (g,) = torch.autograd.grad(total, self.comp)
        return torch.stack([terms[k].detach() for k in self.names]), g

    def _capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                self._compute()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out_terms, self.out_grad = self._compute()

    @torch.no_grad()
    def _load(self, comp, batch):
        r = comp.shape[0]
        assert r <= self.capacity
        self.comp.data[:r].copy_(comp.detach())
        for k in self.KEYS:
            self.bufs[k][:r].copy_(batch[k])
        self.n_rays.fill_(r)
        return r

    def run(self, comp, batch):
        r = self._load(comp, batch)
        if self.graph is None:
            self._capture()
        self.graph.replay()
        terms = {k: self.out_terms[i] for i, k in enumerate(self.names)}
        return terms, self.out_grad[:r]


class TableAdamW:
    """torch.optim.AdamW for the hash-table parameter alone (same update, configs/
    neuralangelo-ortho-wmask.yaml:96-110), one fused launch per step over the ACTIVE levels:
    update + f16 image + gradient reset (csrc/nsr_step.hip table_adamw_kernel).  Levels the
    progressive schedule still masks have zero gradients and zero moments, so their AdamW step is
    `p *= 1 - lr * wd` only; that product is accumulated on the host (float64) and applied when
    the level is switched on and in finalize() — every level ends at the value the dense
    optimizer gives, without 3000 passes over the 80 % of the table nobody reads."""

    def __init__(self, enc, lr, betas, eps, weight_decay=0.01):
        self.enc = enc
        p = enc.params
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.m = torch.zeros_like(p.data)
        self.v = torch.zeros_like(p.data)
        self.grad = torch.zeros_like(p.data)       # persistent gradient buffer, zeroed by the step
        self.step_count = 0
        self.offsets = [2 * o for o in enc.cfg.levels()["offsets"]]      # float offsets per level
        self.n_levels = enc.cfg.n_levels
        self.active = 0                            # levels below this index are up to date
        self.pending = 1.0                         # product of (1 - lr wd) not yet applied to levels >= active
        self.img = None                            # locked at the first step (parameters may still be loaded)

    def _catch_up(self, upto):
        """Apply the pending decay to levels [self.active, upto) and mark them active."""
        if upto > self.active:
            a, b = self.offsets[self.active], self.offsets[upto]
            if self.pending != 1.0 and b > a:
                ops.table_decay(self.enc.params.data, self.img, a, b - a, self.pending)
            self.active = upto

    def activate(self, active_levels):
        """Called when the level schedule moves, BEFORE the step's forward: a level that is
        switched on enters its first forward / backward with its accumulated decay applied, as it
        would under the dense optimizer."""
        if self.img is None or self.enc._shadow is not self.img or not self.enc._shadow_locked:
            self.img = self.enc.lock_shadow()
        self._catch_up(max(active_levels, self.active))

    def prepare_step(self, active_levels, lr=None):
        """Host bookkeeping of one update: returns the launch arguments of `dsu_table_adamw`
        (n floats of the active levels, lr, bias corrections).  `commit_step(lr)` afterwards."""
        lr = self.lr if lr is None else lr
        if self.img is None or self.enc._shadow is not self.img or not self.enc._shadow_locked:
            self.img = self.enc.lock_shadow()      # first step, or someone invalidated the image
        if active_levels < self.active:
            # a level was switched OFF again (not in the reference's schedule): its moments are
            # live, handle everything densely from here on
            active_levels = self.active
        self._catch_up(active_levels)
        b1, b2 = self.betas
        k = self.step_count + 1
        return self.offsets[self.active], lr, 1.0 - b1 ** k, math.sqrt(1.0 - b2 ** k)

    def commit_step(self, lr):
        self.step_count += 1
        self.pending *= 1.0 - lr * self.wd

    def step(self, active_levels, lr=None):
        n, lr, bc1, bc2_sqrt = self.prepare_step(active_levels, lr)
        b1, b2 = self.betas
        ops.table_adamw(self.enc.params.data, self.grad, self.m, self.v, self.img, n, lr, b1, b2,
                        self.eps, self.wd, bc1, bc2_sqrt)
        self.commit_step(lr)

    def finalize(self):
        """Bring the still-masked levels up to date (end of fit, before state_dict / export)."""
        if self.img is None:
            return
        act = self.active
        self._catch_up(self.n_levels)
        self.active = act
        self.pending = 1.0           # the levels >= act are current; later steps accumulate from 1


class SmallAdamW:
    """torch.optim.AdamW (same update, per-group lr, per-tensor step count) for the model's small
    tensors in ONE launch (`dsu_adamw_multi`): torch's fused optimizer spends a step-counter launch
    and an update launch per parameter group.  Tensors without a gradient are skipped, as torch
    does."""

    def __init__(self, torch_optimizer, betas, eps, weight_decay=0.01):
        self.opt = torch_optimizer                    # owns the groups (lr schedule, zero_grad)
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        self.state = {}                               # id(param) -> [m, v, step]
        n = sum(len(g["params"]) for g in self.opt.param_groups)
        assert n <= ops._lib.ADAMW_MAX_TENSORS, "more small tensors than dsu_adamw_multi takes"

    def step(self):
        b1, b2 = self.betas
        entries = []
        for group in self.opt.param_groups:
            lr = group["lr"]
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                st = self.state.get(id(p))
                if st is None:
                    st = self.state[id(p)] = [torch.zeros_like(p.data), torch.zeros_like(p.data), 0]
                st[2] += 1
                if not g.is_contiguous():
                    g = g.contiguous()
                entries.append((p.data, g, st[0], st[1], lr, 1.0 - b1 ** st[2],
                                math.sqrt(1.0 - b2 ** st[2])))
        ops.adamw_multi(entries, b1, b2, self.eps, self.wd)


# HIP-event timing of the native driver's geometry launches (bench.py's roofline object): bench.py
# flips "enabled"; finished drivers add their (launches, ms, algorithmic bytes) to "totals".
# "stride": every n-th step is timed (1 = all; bench.py uses 7: the event records of a timed step
# cost ~28 us of its ~1.05 ms).
native_timing = {"enabled": False, "totals": {}, "stride": 1}


class NativeStepDriver:
    """The optimisation step sequenced by the library itself (csrc/nsr_driver.hip,
    `dsu_nsr_driver_step`): per step the interpreter fills a dozen scalars and makes ONE call; the
    ~20 launches of the step, the side-stream prefetch of the next step's samples, the random
    draws (Philox) and the small tensors' weight-norm / AdamW run from C.  Same kernels and the
    same order as `OrthoNeuSSystem.training_step_fused`; the hash table's AdamW (level
    bookkeeping) and the occupancy refresh of every 16th step stay with the caller."""

    def __init__(self, system):
        import ctypes as C
        from .. import _lib
        self.C, self._lib = C, _lib
        m, ds, dev = system.model, system.dataset, system.device
        geo = m.geometry
        lin0, lin1 = [l for l in geo.network.layers if isinstance(l, torch.nn.Linear)]
        if not (hasattr(lin0, "weight_g") and m.texture.fused_ok and m.fused_shading):
            raise ops.DsuError("the native step driver is built for the reference networks "
                               "(weight-normed SDF MLP, 16-64-64-3 texture MLP)")
        tensors = (ds.all_c2w, ds.origins, ds.directions, ds.all_images, ds.all_normals_world,
                   ds.all_masks, ds.view_weights)
        if not all(t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda for t in tensors):
            raise ops.DsuError("the native step driver needs the dataset resident as contiguous f32")
        L, oc, mc = system.config.loss, system.config.optimizer, m.config
        c = _lib.NsrDriverCfg()
        c.grid = geo.hashgrid.cfg.c()
        c.radius, c.render_step_size = float(geo.radius), float(m.render_step_size)
        c.cap_points, c.cap_rays = _PACK_CAPACITY, int(mc.max_train_num_rays)
        c.n_random, c.sort_bits = 2048, _SPATIAL_SORT_BITS
        c.dynamic_ray_sampling = int(bool(mc.dynamic_ray_sampling))
        c.train_num_samples = int(system.train_num_samples)
        for name, t in zip(("c2w", "origins", "directions", "images", "normals", "masks",
                            "view_weights"), tensors):
            setattr(c, name, t.data_ptr())
        c.V, c.H, c.W, c.image_channels = (int(v) for v in ds.all_images.shape)
        self.params = [lin0.weight_v, lin0.weight_g, lin0.bias, lin1.weight_v, lin1.weight_g,
                       lin1.bias, *m.texture.fused_params(), m.variance.variance]
        assert all(p.is_contiguous() and p.dtype == torch.float32 for p in self.params)
        for name, p_ in zip(("w0_v", "w0_g", "b0", "w1_v", "w1_g", "b1"), self.params[:6]):
            setattr(c, name, p_.data_ptr())
        for k, p_ in enumerate(self.params[6:12]):
            c.tex[k] = p_.data_ptr()
        c.variance = self.params[12].data_ptr()
        c.ray_loss = _lib.RayLossCfg(
            float(L.rgb_p_ratio), float(L.normal_p_ratio), float(L.mask_p_ratio),
            float(L.lambda_rgb_mse), float(L.lambda_rgb_l1 or 0.0), float(L.lambda_normal),
            float(L.lambda_mask if ds.has_mask else 0.0), int(bool(L.geo_aware)), 0)
        c.lambda_eikonal, c.lambda_sparsity = float(L.lambda_eikonal), float(L.lambda_sparsity)
        c.sparsity_scale = float(L.sparsity_scale)
        c.lambda_smooth = float(L.lambda_3d_normal_smooth) if L.lambda_3d_normal_smooth > 0 else 0.0
        c.beta1, c.beta2 = float(oc.betas[0]), float(oc.betas[1])
        c.adam_eps, c.weight_decay = float(oc.eps), 0.01
        c.seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        need = int(_lib.lib().dsu_nsr_driver_workspace_bytes(C.byref(c)))
        if need < 0:
            ops.check(need, "dsu_nsr_driver_workspace_bytes")
        self.workspace = torch.empty(need, dtype=torch.uint8, device=dev)
        c.workspace, c.workspace_bytes = self.workspace.data_ptr(), need
        self.cfg = c
        self.keep = tensors                                   # the driver holds raw pointers
        h = C.c_void_p()
        ops.check(_lib.lib().dsu_nsr_driver_create(C.byref(c), C.byref(h)), "dsu_nsr_driver_create")
        self.handle = h
        self.args = _lib.NsrStepArgs()
        self.adam_step = 0
        self.param_versions = [p._version for p in self.params]
        off = int(_lib.lib().dsu_nsr_driver_terms(h)) - self.workspace.data_ptr()
        self.terms2 = self.workspace[off:off + 64].view(torch.float32).view(2, 8)
        self.timing_on = False
        self.dataset = ds
        import weakref
        self._system = weakref.ref(system)                    # no cycle: the system owns the driver
        self.stepped = False                                  # effective weights exist after a step
        self._occ_ws = None
        if getattr(mc, "grid_prune", False):
            m.occupancy_grid.native_refresh = self.occ_refresh

    def set_timing(self, on):
        """on: 0 / False = off, n = time every n-th step."""
        on = int(on)
        if on != self.timing_on:
            if self.timing_on:
                self.flush_timing()
            ops.check(self._lib.lib().dsu_nsr_driver_timing(self.handle, on), "dsu_nsr_driver_timing")
            self.timing_on = on

    def flush_timing(self):
        """Add what has been timed so far to native_timing["totals"] and start over."""
        if not self.timing_on:
            return
        C = self.C
        for fam, name in ((0, "sdf_fd_fwd"), (1, "sdf_fd_bwd")):
            n, ms, work = C.c_int64(), C.c_double(), C.c_double()
            ops.check(self._lib.lib().dsu_nsr_driver_timing_read(
                self.handle, fam, C.byref(n), C.byref(ms), C.byref(work)), "dsu_nsr_driver_timing_read")
            fl = C.c_double()
            ops.check(self._lib.lib().dsu_nsr_driver_timing_flops(self.handle, fam, C.byref(fl)),
                      "dsu_nsr_driver_timing_flops")
            t = native_timing["totals"].setdefault(name, [0, 0.0, 0.0, 0.0])
            t[0] += n.value; t[1] += ms.value; t[2] += work.value; t[3] += fl.value
        ops.check(self._lib.lib().dsu_nsr_driver_timing(self.handle, int(self.timing_on)),
                  "dsu_nsr_driver_timing")

    @property
    def system(self):
        return self._system()

    def occ_refresh(self, grid, step, all_cells, occ_thre, ema_decay, inj_cells=None, inj_rand=None,
                    export=None):
        """OccupancyGrid._update through dsu_nsr_driver_occ_refresh: selection of the cells, points,
        SDF with the driver's effective weights, alpha, EMA, mean, binarisation — on the current
        stream, no host round trip.  Returns False (the caller's torch path runs) before the
        driver's first step, when its effective weights do not exist yet.  export: a dict that
        receives "cells" (int32, -1 = unused slot) and "rand" (x 3 uniforms) of the call (tests)."""
        C, _lib = self.C, self._lib
        if self.handle is None or not self.stepped or self.system is None:
            return False
        if [p._version for p in self.params] != self.param_versions:
            return False          # parameters written from outside: the driver's weights are stale
        geo = self.system.model.geometry
        dev = grid.occs.device
        need = int(_lib.lib().dsu_occgrid_refresh_workspace_bytes(grid.res))
        if self._occ_ws is None or self._occ_ws.numel() < need:
            self._occ_ws = torch.empty(need, dtype=torch.uint8, device=dev)
        binary = grid.binary_u8()
        a = _lib.OccRefreshArgs()
        a.occs, a.binary = grid.occs.data_ptr(), binary.data_ptr()
        a.res, a.all_cells = int(grid.res), int(bool(all_cells))
        a.step = int(step)
        a.table_img = geo.hashgrid.table_f16().data_ptr()
        a.active_levels = int(geo.active_levels)
        a.ema_decay, a.occ_thre = float(ema_decay), float(occ_thre)
        keep = []
        if inj_cells is not None:
            ic = inj_cells.to(dev, torch.int32).contiguous()
            ir = inj_rand.to(dev, torch.float32).contiguous()
            keep += [ic, ir]
            a.inj_count, a.inj_cells, a.inj_rand = int(ic.numel()), ic.data_ptr(), ir.data_ptr()
        if export is not None:
            m = int(ic.numel()) if inj_cells is not None else \
                (grid.num_cells if all_cells else 2 * (grid.num_cells // 4))
            export["cells"] = torch.empty(m, dtype=torch.int32, device=dev)
            export["rand"] = torch.empty(m, 3, dtype=torch.float32, device=dev)
            a.cells_out, a.rand_out = export["cells"].data_ptr(), export["rand"].data_ptr()
        a.workspace, a.workspace_bytes = self._occ_ws.data_ptr(), need
        ops.check(_lib.lib().dsu_nsr_driver_occ_refresh(self.handle, C.byref(a), ops.stream()),
                  "dsu_nsr_driver_occ_refresh")
        grid._binary_u8 = binary
        grid._binary = binary.view(grid.res, grid.res, grid.res).bool()
        return True

    def close(self):
        grid = getattr(self.system.model, "occupancy_grid", None) if self.system is not None else None
        if grid is not None and getattr(grid, "native_refresh", None) == self.occ_refresh:
            grid.native_refresh = None
        if self.handle is not None:
            self.flush_timing()
            self._lib.lib().dsu_nsr_driver_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OrthoNeuSSystem:
    def __init__(self, model_config=None, system_config=None, device="cuda", seed=123456):
        torch.manual_seed(seed)
        np.random.seed(seed)
        self.device = torch.device(device)
        self.model = NeuSModel(model_config).to(self.device)
        self.config = Cfg(system_config) if system_config is not None else DEFAULT_SYSTEM_CONFIG
        mc = self.model.config
        self.train_num_samples = mc.train_num_rays * mc.num_samples_per_ray
        self.train_num_rays = mc.train_num_rays
        self.global_step = 0
        oc = self.config.optimizer
        # the hash table has its own fused optimizer step (TableAdamW, same AdamW update);
        enc = self.model.geometry.hashgrid
        self.table_opt = None
        self.keep_table_grad = False
        if self.device.type == "cuda":
            self.table_opt = TableAdamW(enc, dict(oc.params)["geometry"], tuple(oc.betas), oc.eps)
            # whoever reads the parameters (checkpoint, export) sees the masked levels with their
            # lazily accumulated weight decay applied
            self.model.register_state_dict_pre_hook(lambda *a, **k: self.table_opt.finalize())
        groups = [{"params": [p for p in getattr(self.model, n).parameters()
                              if self.table_opt is None or p is not enc.params],
                   "name": n, "lr": lr}
                  for n, lr in oc.params.items()]
        # same update rule as the reference's torch.optim.AdamW; the fused (single-launch per
        # group) implementation on the GPU instead of ~10 foreach launches per group
        fused = self.device.type == "cuda"
        self.optimizer = torch.optim.AdamW(groups, lr=oc.lr, betas=tuple(oc.betas), eps=oc.eps,
                                           **({"fused": True} if fused else {}))
        self._base_lrs = [g["lr"] for g in groups]
        # the small tensors' AdamW as one launch
        self.small_opt = None
        if self.device.type == "cuda":
            self.small_opt = SmallAdamW(self.optimizer, tuple(oc.betas), oc.eps)
        # ExponentialLR gamma = 0.1 ** (1 / (max_steps - constant_steps))  (recon.py:13)
        # (a run that ends inside the constant phase never reaches the decay: factor 1)
        decay_steps = self.config.max_steps - self.config.constant_steps
        self._gamma = 0.1 ** (1.0 / decay_steps) if decay_steps > 0 else 1.0
        self.dataset = None
        self.last = {}
        self.use_loss_graph = self.device.type == "cuda"
        self._loss_graph = None
        # "fused": forward and backward of a step driven kernel by kernel (no autograd graph
        # except for the texture MLP and the tiny weight-norm / variance chains);
        # "autograd": the op-by-op step through torch.autograd (cross-check, same numbers)
        # "native": the step sequenced by the library (NativeStepDriver) whenever no draws are
        # injected; "fused": the same kernels sequenced from here
        self.step_mode = "native"
        self.native_with_injected_draws = False     # tests: injected draws through the native driver
        self._native, self._python_steps = None, 0
        self.use_prefetch = self.device.type == "cuda"
        self.fused_batch = True
        self._side, self._prefetched = None, None
        self.pack_on_side_stream = True
        self._packed = {}
        self._stats_pinned = [torch.empty(2, dtype=torch.int32).pin_memory() for _ in range(2)] \
            if self.device.type == "cuda" and torch.cuda.is_available() else None

    # ----------------------------------------------------------------- data
    def preprocess_data(self, index=None, x=None, y=None):
        ds, n = self.dataset, self.train_num_rays
        dev = self.device
        if index is None:
            index = torch.randint(0, len(ds.all_masks), size=(n,), device=dev)
            x = torch.randint(0, ds.w, size=(n,), device=dev)
            y = torch.randint(0, ds.h, size=(n,), device=dev)
        if self.device.type == "cuda" and self.fused_batch and ds.all_images.dtype == torch.float32 \
                and all(t.is_contiguous() for t in (ds.all_images, ds.all_normals_world,
                                                    ds.all_masks, ds.view_weights, ds.origins,
                                                    ds.directions, ds.all_c2w)):
            # one launch instead of ~30 (gathers, two batched 3x3 products, norms, cat)
            return ops.ortho_ray_batch(index, x, y, ds.all_c2w, ds.origins, ds.directions,
                                       ds.all_images, ds.all_normals_world, ds.all_masks,
                                       ds.view_weights)
        return self.preprocess_data_torch(index, x, y)

    def preprocess_data_torch(self, index, x, y):
        """The op-by-op form of preprocess_data (cross-check for the fused kernel)."""
        ds = self.dataset
        c2w = ds.all_c2w[index]
        directions, origins = ds.directions[index, y, x], ds.origins[index, y, x]
        rays_o, rays_d = get_ortho_rays(origins, directions, c2w)
        rgb = ds.all_images[index, y, x].view(-1, ds.all_images.shape[-1])
        normal = ds.all_normals_world[index, y, x].view(-1, 3)
        mask = ds.all_masks[index, y, x].view(-1)
        view_weights = ds.view_weights[index, y, x].view(-1)
        cosines = F.cosine_similarity(rays_d, normal, dim=-1, eps=1e-6)
        rays = torch.cat([rays_o, F.normalize(rays_d, p=2, dim=-1)], -1)
        return {"rays": rays, "rgb": rgb, "normal": normal, "mask": mask, "cosines": cosines,
                "view_weights": view_weights}

    # ----------------------------------------------------------------- losses
    def ray_losses(self, comp, batch, pad=None):
        """The three ray-level terms (neus_ortho.py:94-133) from the raw composite
        comp (R,8) = [opacity, depth, rgb(3), sum w*normal(3)].  `pad` (R,) bool marks the real
        rays when the tensors are padded to a fixed capacity (graph replay); None = all real."""
        L = self.config.loss
        comp_rgb, opacity = comp[:, 2:5], comp[:, 0]
        comp_normal = F.normalize(comp[:, 5:8], p=2, dim=-1)
        view_weights = batch["view_weights"]
        cosines = torch.where(batch["cosines"] > -0.1, torch.zeros_like(batch["cosines"]),
                              batch["cosines"])                         # cosines[cosines > -0.1] = 0
        mask = (batch["mask"] > 0) & (cosines < -0.1)
        real = torch.ones_like(mask) if pad is None else pad
        mask = mask & real
        terms = {}
        # x[mask] of the reference evaluated without host round trips (ranking_loss_masked)
        err = F.mse_loss(comp_rgb, batch["rgb"], reduction="none")
        terms["rgb_mse"] = ranking_loss_masked(err.sum(1), mask, L.rgb_p_ratio, type="mean") \
            * L.lambda_rgb_mse
        if L.lambda_rgb_l1:
            l1 = F.l1_loss(comp_rgb, batch["rgb"], reduction="none")
            terms["rgb_l1"] = ranking_loss_masked(l1.sum(1), mask, L.rgb_p_ratio) * L.lambda_rgb_l1
        normal_errors = 1 - F.cosine_similarity(comp_normal, batch["normal"], dim=1)
        if L.geo_aware:
            e = torch.exp(cosines.abs())
            normal_errors = normal_errors * e / (e * real).sum()
            ln = ranking_loss_masked(normal_errors, mask, L.normal_p_ratio, view_weights, "sum")
        else:
            ln = ranking_loss_masked(normal_errors, mask, L.normal_p_ratio, view_weights, "mean")
        terms["normal"] = ln * L.lambda_normal
        opac = torch.clamp(opacity, 1e-3, 1 - 1e-3)
        lm = ranking_loss_masked(binary_cross_entropy(opac, batch["mask"].float()), real,
                                 L.mask_p_ratio, view_weights)
        terms["mask"] = lm * (L.lambda_mask if self.dataset.has_mask else 0.0)
        return terms

    def sample_losses(self, out):
        """Sample-level terms: eikonal, sparsity, 3-D normal smoothness (neus_ortho.py:118-151)."""
        L = self.config.loss
        terms = {}
        terms["eikonal"] = ((torch.linalg.norm(out["sdf_grad_samples"], ord=2, dim=-1) - 1.0) ** 2
                            ).mean() * L.lambda_eikonal
        terms["sparsity"] = torch.exp(-L.sparsity_scale * out["random_sdf"].abs()).mean() \
            * L.lambda_sparsity
        if L.lambda_3d_normal_smooth > 0:
            terms["normal_smooth"] = (out["random_sdf_grad"] - out["normal_perturb"]).abs().mean() \
                * L.lambda_3d_normal_smooth
        return terms

    def losses(self, out, batch):
        comp = out.get("comp_raw")
        if comp is None:      # op-by-op path: rebuild the raw composite layout
            comp = torch.cat([out["opacity"], out["depth"], out["comp_rgb"], out["comp_normal"]], 1)
        return {**self.ray_losses(comp, batch), **self.sample_losses(out)}

    # ----------------------------------------------------------------- one optimisation step
    def _set_lr(self):
        s = self.global_step
        f = 1.0 if s < self.config.constant_steps else self._gamma ** (s - self.config.constant_steps)
        for g, base in zip(self.optimizer.param_groups, self._base_lrs):
            g["lr"] = base * f

    def zero_grad(self):
        self.optimizer.zero_grad(set_to_none=True)
        self.model.geometry.hashgrid.params.grad = None     # not in the torch optimizer's groups

    def _step_small(self):
        if self.small_opt is not None:
            self.small_opt.step()
        else:
            self.optimizer.step()

    def _step_table(self, active_levels):
        """Optimizer step of the hash table: the fused kernel (gradient in table_opt.grad), or —
        when the table sits in the torch optimizer — just the image invalidation (torch's fused
        AdamW does not bump ._version)."""
        if self.table_opt is None:
            self.model.geometry.hashgrid.invalidate()
            return
        lr = next(g["lr"] for g in self.optimizer.param_groups if g.get("name") == "geometry")
        if self.keep_table_grad:              # tests: the step's table gradient stays readable
            self.model.geometry.hashgrid.params.grad = self.table_opt.grad.clone()
        self.table_opt.step(int(active_levels), lr)

    def training_step(self, inject=None):
        fusable = self.model.fused_shading and self.train_num_rays <= ops.RAY_LOSS_MAX_RAYS
        native_ok = (self.step_mode == "native" and fusable and self.device.type == "cuda"
                     and self.table_opt is not None and self.use_prefetch
                     and self.model.config.max_train_num_rays <= ops.RAY_LOSS_MAX_RAYS)
        # The two sequencings keep separate optimizer moments for the small tensors, so a system
        # stays on the one it started with: native when its first step draws its own rays,
        # Python-sequenced when the first step comes with injected draws (tests of that path).
        if native_ok and (self._native is not None
                          or (self._python_steps == 0 and not (inject and "batch" in inject)
                              and (not inject or self.native_with_injected_draws))):
            return self.training_step_native(inject)
        self._python_steps += 1
        if self.step_mode in ("fused", "native") and fusable:
            return self.training_step_fused(inject)
        return self.training_step_autograd(inject)

    def training_step_native(self, inject=None):
        """One optimisation step through `dsu_nsr_driver_step` (see NativeStepDriver)."""
        import ctypes as C
        m = self.model
        geo, enc = m.geometry, m.geometry.hashgrid
        m.train()
        m.update_step(0, self.global_step)             # level / eps schedule, occupancy refresh
        drv = self._native
        if drv is None or drv.dataset is not self.dataset:
            if drv is not None:
                drv.close()
            drv = self._native = NativeStepDriver(self)
        drv.set_timing(int(native_timing.get("stride", 1)) if native_timing["enabled"] else 0)
        a = drv.args
        inject = inject or {}
        keep = []
        # tests may hand over a whole ray batch (rays, rgb, normal, mask, cosines, view_weights), as
        # to training_step_fused: copied into the driver's sample set instead of the dataset gathers
        batch = inject.get("batch") or {}
        for field, key in (("inj_rays", "rays"), ("inj_rgb", "rgb"), ("inj_normal", "normal"),
                           ("inj_mask", "mask"), ("inj_cosines", "cosines"),
                           ("inj_view_weights", "view_weights")):
            t = batch.get(key)
            if t is not None:
                t = t.to(self.device, torch.float32).contiguous()
                keep.append(t)
            setattr(a, field, None if t is None else t.data_ptr())
        if batch and int(batch["rays"].shape[0]) != int(self.train_num_rays):
            raise ValueError("injected ray batch: train_num_rays must equal its row count")
        for field, key, dt in (("inj_index", "index", torch.int64), ("inj_x", "x", torch.int64),
                               ("inj_y", "y", torch.int64), ("inj_jitter", "jitter", torch.float32),
                               ("inj_pts_random", "pts_random", torch.float32),
                               ("inj_perturb", "perturb", torch.float32)):
            t = inject.get(key)
            if t is not None:
                t = t.to(self.device, dt).contiguous()
                keep.append(t)
            setattr(a, field, None if t is None else t.data_ptr())
        nxt = self.global_step + 1
        a.step, a.n_rays = self.global_step, int(self.train_num_rays)
        a.prefetch_next = int(not inject and not (m.config.grid_prune and nxt % 16 == 0))
        a.active_levels, a.eps = int(geo.active_levels), float(geo._finite_difference_eps)
        a.cos_anneal_ratio = float(m.cos_anneal_ratio)
        self._set_lr()
        lrs = {g.get("name"): g["lr"] for g in self.optimizer.param_groups}
        a.lr_geometry, a.lr_texture, a.lr_variance = lrs["geometry"], lrs["texture"], lrs["variance"]
        a.adam_step = drv.adam_step + 1
        a.randomized = int(bool(m.randomized))
        versions = [p._version for p in drv.params]    # load_state_dict & co. bump these
        a.refresh_effective = int(versions != drv.param_versions)
        drv.param_versions = versions
        if m.config.grid_prune:
            occ = m.occupancy_grid.binary_u8()
            a.occ_binary, a.occ_res = occ.data_ptr(), m.occupancy_grid.res
        else:
            a.occ_binary, a.occ_res = None, 0
        topt = self.table_opt
        topt.activate(int(geo.active_levels))
        n_tab, lr_tab, bc1, bc2s = topt.prepare_step(int(geo.active_levels), lrs["geometry"])
        a.table_img, a.table_grad = topt.img.data_ptr(), topt.grad.data_ptr()
        a.table_p, a.table_m, a.table_v = enc.params.data_ptr(), topt.m.data_ptr(), topt.v.data_ptr()
        if self.keep_table_grad:              # tests: no table update, its gradient stays in topt.grad
            a.table_p = None
        a.table_n, a.table_lr, a.table_bc1, a.table_bc2_sqrt = int(n_tab), lr_tab, bc1, bc2s
        a.table_eps, a.table_wd = float(topt.eps), float(topt.wd)
        # the step's loss terms come back as a copy in fresh memory (the driver's two sets are reused
        # two steps later): a result kept across steps must not change under its holder
        terms_copy = torch.empty(8, dtype=torch.float32, device=self.device)
        a.terms_out = terms_copy.data_ptr()
        for attempt in range(6):
            rc = drv._lib.lib().dsu_nsr_driver_step(drv.handle, C.byref(a), ops.stream())
            if rc != -3 or a.out_n_samples <= _PACK_CAPACITY or a.n_rays <= 64:
                break
            # more samples than the packed buffers hold (twice the schedule's target): nothing of
            # the step has run yet — march half the rays instead (the dynamic ray schedule would
            # have shrunk the batch over the next steps anyway)
            a.n_rays = max(64, a.n_rays // 2)
            self.train_num_rays = int(a.n_rays)
        if rc != 0:
            raise ops.DsuError(f"dsu_nsr_driver_step failed ({rc}): {a.out_n_samples} samples, "
                               f"longest ray {a.out_max_count} (capacity {_PACK_CAPACITY})")
        drv.adam_step += 1
        drv.stepped = True
        n_rays = int(self.train_num_rays)
        if m.config.dynamic_ray_sampling:
            self.train_num_rays = int(a.out_next_n_rays)
        if self.keep_table_grad:
            enc.params.grad = topt.grad.clone()
        else:
            topt.commit_step(lr_tab)           # the update itself was launched by the driver
        self.global_step += 1
        t = terms_copy
        L = self.config.loss
        terms = {"rgb_mse": t[0]}
        if L.lambda_rgb_l1:
            terms["rgb_l1"] = t[1]
        terms.update({"normal": t[2], "mask": t[3], "eikonal": t[4], "sparsity": t[5]})
        if L.lambda_3d_normal_smooth > 0:
            terms["normal_smooth"] = t[6]
        self.last = {"loss": _LazyLoss(t[0:4], t[4:7]), "n_samples": int(a.out_n_samples),
                     "n_rays": n_rays, **terms}
        return self.last

    # ------------------------------------------------------------- ray batch + march (prefetchable)
    def _march_begin(self, inject=None):
        """preprocess_data + ray/box intersection + single-pass march + offsets scan on the
        CURRENT stream; nothing waits for the device."""
        m = self.model
        inject = inject or {}
        # tests may hand over a whole ray batch (rays, rgb, normal, mask, cosines, view_weights)
        batch = inject.get("batch") or \
            self.preprocess_data(inject.get("index"), inject.get("x"), inject.get("y"))
        rays = batch["rays"]
        rays_o, rays_d = rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous()
        jitter = inject.get("jitter")
        if jitter is None and m.randomized:
            jitter = torch.rand(rays.shape[0], device=rays.device)
        with torch.no_grad():
            tmin, tmax = ops.ray_aabb(rays_o, rays_d, m._aabb_host,
                                      jitter if m.randomized else None, m.render_step_size)
            occ, res = None, 0
            if m.config.grid_prune:
                occ, res = m.occupancy_grid.binary_u8(), m.occupancy_grid.res
            h = ops.ray_march_begin(rays_o, rays_d, tmin, tmax, m._aabb_host, occ, res,
                                    m.render_step_size)
        # the 2048 random points of the sparsity / smoothness terms and their perturbation
        # (neus.py:155-162) are independent of the parameters as well
        pts_random, perturb = inject.get("pts_random"), inject.get("perturb")
        if pts_random is None:
            pts_random = torch.rand([1024 * 2, 3], device=rays.device) * 2 - 1
        if perturb is None:
            perturb = torch.randn_like(pts_random)
        return {"batch": batch, "handle": h, "keep": (tmin, tmax, jitter, rays),
                "pts_random": pts_random, "perturb": perturb}

    def _launch_prefetch(self, done=None):
        """The next step's ray batch and march depend on the occupancy grid and the RNG, not on
        the parameters this step is about to update: run them on a side stream, concurrently with
        this step's geometry / backward kernels (32 marching waves next to a 4-wave-per-CU
        backward), and hand the sample count to the host through pinned memory.  Skipped when
        the next step refreshes the occupancy grid (every 16th step) — that must see the updated
        parameters first."""
        if not self.use_prefetch:
            return
        nxt = self.global_step + 1
        if self.model.config.grid_prune and nxt % 16 == 0:
            return
        main = torch.cuda.current_stream()
        if self._side is None:
            # high priority: its 32 marching waves must not queue behind the main stream's
            # chip-filling kernels, or the next step blocks on the sample count
            self._side = torch.cuda.Stream(priority=-1)
        if done is None:
            done = torch.cuda.Event()
            done.record(main)
        self._side.wait_event(done)            # compaction of THIS step (and a grid refresh) done
        with torch.cuda.stream(self._side):
            prep = self._march_begin()
            if self.pack_on_side_stream:
                # compaction + random tail + Morton sort of the next step's points, off the main
                # stream (8 launches, ~55 us per step)
                n_r = prep["pts_random"].shape[0]
                key = nxt & 1
                bufs = self._packed.get(key)
                if bufs is None or bufs.tail_rows != 2 * n_r:
                    bufs = self._packed[key] = ops.PackedStepBuffers(
                        _PACK_CAPACITY, 2 * n_r, _SPATIAL_SORT_BITS, self.device)
                ops.ray_pack_prefetched(prep["handle"], bufs, prep["pts_random"], prep["perturb"],
                                        self.model.geometry.radius)
                prep["packed"] = bufs
            host = self._stats_pinned[nxt & 1]
            host.copy_(prep["handle"].stats, non_blocking=True)
            prep["stats_host"] = host
            prep["ready"] = torch.cuda.Event()
            prep["ready"].record(self._side)
        prep["step"] = nxt
        # No tensor.record_stream(main) here.  These tensors come from the side stream's pool and
        # are read by main-stream kernels of the NEXT step; their blocks return to that pool when
        # the step drops them and can only be re-used by a later prefetch, which the side stream
        # starts after waiting for an event recorded on main behind that step's kernels (`done`
        # above), so the re-use is already ordered.  record_stream would make the allocator put
        # one hipEventRecord per tensor (16) into the MAIN queue between AdamW and the next step's
        # compaction: ~80 us of command-processor work per step (hip-runtime trace).
        self._prefetched = prep

    def _take_prefetch(self):
        prep, self._prefetched = self._prefetched, None
        if prep is not None and (prep["step"] != self.global_step
                                 or prep["handle"].n != self.train_num_rays):
            return None
        return prep

    def training_step_fused(self, inject=None):
        """Same step as training_step_autograd (neus_ortho.py:84-160 + the model forward
        neus.py:115-196), with the backward pass sequenced by hand over the fused kernels:
        march(+positions) -> geometry (7-eval fused) -> shading prep -> texture MLP -> compositing
        -> ray losses (+d/d comp) -> compositing bwd -> texture bwd -> shading bwd -> sample
        losses (+gradients) -> geometry bwd -> weight-norm / variance chains -> AdamW."""
        from .render import RayPacking
        m, L = self.model, self.config.loss
        geo, enc = m.geometry, m.geometry.hashgrid
        m.train()
        inject = inject or {}
        m.update_step(0, self.global_step)
        if self.table_opt is not None:
            self.table_opt.activate(int(geo.active_levels))
        prep = self._take_prefetch() if not inject else None
        if prep is None:
            prep = self._march_begin(inject)
            total, cmax = prep["handle"].stats.tolist()              # the step's one host sync
        else:
            # The host waits for the event, so everything the side stream did before it is
            # complete: no device-side wait is needed on the main stream (a cross-queue barrier
            # packet cost ~80 us of main-stream idle time per step even when already signalled).
            prep["ready"].synchronize()                              # stats are in pinned memory
            total, cmax = prep["stats_host"].tolist()
        batch, rays_d, h = prep["batch"], prep["handle"].rays_d, prep["handle"]
        dev = rays_d.device
        n_rays = rays_d.shape[0]
        pts_random, perturb = prep["pts_random"], prep["perturb"]
        n_r = pts_random.shape[0]
        packed = prep.get("packed")
        perm, presorted = None, False
        off, cnt, n_s = h.offsets, h.counts, total
        RayPacking.total = n_s
        if packed is not None and cmax <= h.cap and total <= packed.capacity:
            # the side stream already packed the samples, appended the random points and sorted
            # everything (fixed-capacity buffers; the total was on the device only)
            n_pk = n_s + 2 * n_r
            ts, te = packed.t_starts[:n_s], packed.t_ends[:n_s]
            if packed.bits:
                allp, perm, presorted = packed.sorted[:n_pk], packed.perm[:n_pk], True
            else:
                allp = packed.points[:n_pk]
        else:
            with torch.no_grad():
                allp, ts, te = ops.ray_march_finish(h, total, cmax, tail_rows=2 * n_r)
                allp[n_s:n_s + n_r] = pts_random
                torch.add(pts_random, perturb, alpha=1e-2, out=allp[n_s + n_r:])
        # the march scratch rows may be overwritten by the next prefetch once this has run
        done_event = torch.cuda.Event()
        done_event.record(torch.cuda.current_stream())
        if m.config.dynamic_ray_sampling and n_s > 0:
            tr = int(self.train_num_rays * (self.train_num_samples / n_s))
            self.train_num_rays = min(int(self.train_num_rays * 0.9 + tr * 0.1),
                                      m.config.max_train_num_rays)
        n_all = n_s + 2 * n_r
        self._set_lr()
        self.zero_grad()

        # ---- forward
        lin0, lin1 = [l for l in geo.network.layers if isinstance(l, torch.nn.Linear)]
        wn = hasattr(lin0, "weight_g")
        with torch.no_grad():
            if wn:      # weight norm through the ATen ops directly (no autograd graph to walk)
                w0, n0 = torch._weight_norm_interface(lin0.weight_v, lin0.weight_g, 0)
                w1, n1 = torch._weight_norm_interface(lin1.weight_v, lin1.weight_g, 0)
            else:
                w0, w1 = lin0.weight, lin1.weight
        b0, b1 = lin0.bias, lin1.bias
        mlp = [t.detach().contiguous() for t in (w0, b0, w1, b1)]
        table = enc.table_f16()
        eps, active = geo._finite_difference_eps, geo.active_levels
        with torch.no_grad():
            inv_s = m.variance.inv_s                     # exp(10 * variance)
            # the geometry network is evaluated in Morton order of the sample positions (the
            # ray-major order of random pixels scatters a wave's table accesses over the whole
            # volume); its per-point results come back in ray order through `perm`
            if _SPATIAL_SORT_BITS and not presorted:
                allp, perm = ops.spatial_sort(allp, geo.radius, _SPATIAL_SORT_BITS)
            a_sdf, a_grad, a_feat, _, enc_cache = ops.sdf_fd_fwd(
                enc.cfg, table, mlp, allp, geo.radius, eps, active, True, True, False,
                enc_cache=True, perm=perm)
            if not inject:
                # issued once the geometry forward (0.3 ms of device work) is in the queue: the
                # host's ~0.15 ms of launches for the next batch then cost the device nothing, and
                # the march has the rest of this step to finish, so the next step never blocks
                # on it (a cProfile of the loop showed 0.43 ms per step in Event.synchronize when
                # the prefetch was issued behind the geometry backward)
                self._launch_prefetch(done_event)
            normal, tex_in = ops.shade_prep_fwd(a_grad[:n_s], a_feat[:n_s])
        tex_fused = m.texture.fused_ok
        if tex_fused:
            tex_params = [p.detach() for p in m.texture.fused_params()]
            rgb_d = ops.texture_fwd(tex_params, tex_in)
        else:
            tex_in.requires_grad_(True)
            rgb = torch.sigmoid(m.texture.mlp_split_k(tex_in))
            rgb_d = rgb.detach()
        with torch.no_grad():
            inv_d = inv_s.detach().reshape(1)
            car = float(m.cos_anneal_ratio)
            comp, alpha, w = ops.neus_composite_fwd(a_sdf[:n_s], normal, rgb_d, rays_d, ts, te, off,
                                                    cnt, inv_d, car)
            rterms, d_comp = ops.ray_losses(
                comp, batch["rgb"], batch["normal"], batch["mask"], batch["cosines"],
                batch["view_weights"],
                {"rgb_p_ratio": L.rgb_p_ratio, "normal_p_ratio": L.normal_p_ratio,
                 "mask_p_ratio": L.mask_p_ratio, "lambda_rgb_mse": L.lambda_rgb_mse,
                 "lambda_rgb_l1": L.lambda_rgb_l1 or 0.0, "lambda_normal": L.lambda_normal,
                 "lambda_mask": L.lambda_mask if self.dataset.has_mask else 0.0,
                 "geo_aware": L.geo_aware})
            # ---- backward
            d_sdf_all = torch.empty(n_all, device=dev)
            d_grad_all = torch.empty(n_all, 3, device=dev)
            d_feat_all = torch.empty(n_all, 13, device=dev)
            d_feat_all[n_s:].zero_()
            _, d_normal, d_rgb, d_inv = ops.neus_composite_bwd(
                a_sdf[:n_s], normal, rgb_d, rays_d, ts, te, off, cnt, inv_d, car, alpha, w, d_comp,
                None, d_sdf_out=d_sdf_all[:n_s])
            if tex_fused:
                d_tex_in, g_tex = ops.texture_bwd(tex_params, tex_in, rgb_d, d_rgb)
        # d inv_s / d variance = 10 * inv_s
        m.variance.variance.grad = (d_inv.view_as(inv_s) * inv_s * 10.0).view_as(m.variance.variance)
        if tex_fused:
            for p_, g_ in zip(m.texture.fused_params(), g_tex):
                p_.grad = g_
        else:
            torch.autograd.backward([rgb], [d_rgb])
            d_tex_in = tex_in.grad
        with torch.no_grad():
            ops.shade_prep_bwd(a_grad[:n_s], d_normal, d_tex_in,
                               out=(d_grad_all[:n_s], d_feat_all[:n_s]))
            smooth = L.lambda_3d_normal_smooth if L.lambda_3d_normal_smooth > 0 else 0.0
            sterms, _, _ = ops.sample_losses(a_sdf, a_grad, n_s, n_r, L.lambda_eikonal,
                                             L.lambda_sparsity, L.sparsity_scale, smooth,
                                             d_sdf_all, d_grad_all)
            g_table, g = ops.sdf_fd_bwd(enc.cfg, table, mlp, allp, geo.radius, eps, active,
                                        d_sdf_all, d_grad_all, d_feat_all, None,
                                        grad_table=None if self.table_opt is None
                                        else self.table_opt.grad,
                                        enc_cache=enc_cache, perm=perm)
        if self.table_opt is None:
            enc.params.grad = g_table
        if wn:
            with torch.no_grad():
                bw = torch.ops.aten._weight_norm_interface_backward
                lin0.weight_v.grad, lin0.weight_g.grad = bw(g[0], lin0.weight_v, lin0.weight_g, n0, 0)
                lin1.weight_v.grad, lin1.weight_g.grad = bw(g[2], lin1.weight_v, lin1.weight_g, n1, 0)
        else:                                            # no weight norm: w is the parameter
            lin0.weight.grad, lin1.weight.grad = g[0], g[2]
        lin0.bias.grad, lin1.bias.grad = g[1], g[3]
        self._step_small()
        self._step_table(active)
        self.global_step += 1
        terms = {"rgb_mse": rterms[0]}
        if L.lambda_rgb_l1:
            terms["rgb_l1"] = rterms[1]
        terms.update({"normal": rterms[2], "mask": rterms[3], "eikonal": sterms[0],
                      "sparsity": sterms[1]})
        if L.lambda_3d_normal_smooth > 0:
            terms["normal_smooth"] = sterms[2]
        self.last = {"loss": _LazyLoss(rterms, sterms), "n_samples": n_s, "n_rays": n_rays, **terms}
        return self.last

    def training_step_autograd(self, inject=None):
        """One step.  `inject` (tests) = dict(index,x,y,jitter,pts_random,perturb) replaces the
        device RNG draws so that a step is reproducible against the oracle."""
        self.model.train()
        inject = inject or {}
        batch = self.preprocess_data(inject.get("index"), inject.get("x"), inject.get("y"))
        self.model.update_step(0, self.global_step)
        out = self.model(batch["rays"], jitter=inject.get("jitter"),
                         pts_random=inject.get("pts_random"), perturb=inject.get("perturb"))
        # neus_ortho.py:91 reads num_samples back with .item(); the marcher already had to
        # bring the total to the host to size its outputs, so reuse that value (no second sync)
        from .render import RayPacking
        n_samples = RayPacking.total
        if self.model.config.dynamic_ray_sampling and n_samples > 0:
            tr = int(self.train_num_rays * (self.train_num_samples / n_samples))
            self.train_num_rays = min(int(self.train_num_rays * 0.9 + tr * 0.1),
                                      self.model.config.max_train_num_rays)
        self._set_lr()
        self.zero_grad()
        if self.use_loss_graph and "comp_raw" in out:
            # ray-level losses + their gradient w.r.t. the composite: one captured HIP graph
            # (fixed max_train_num_rays capacity, real rays marked by a device-side count)
            if self._loss_graph is None:
                self._loss_graph = RayLossGraph(self, self.model.config.max_train_num_rays)
            rterms, d_comp = self._loss_graph.run(out["comp_raw"], batch)
            sterms = self.sample_losses(out)
            torch.autograd.backward([sum(sterms.values()), out["comp_raw"]], [None, d_comp])
            terms = {**rterms, **sterms}
            loss = sum(terms.values())
        else:
            terms = self.losses(out, batch)
            loss = sum(terms.values())
            loss.backward()
        self._step_small()
        enc = self.model.geometry.hashgrid
        if self.table_opt is not None:
            if enc.params.grad is not None:
                self.table_opt.grad.copy_(enc.params.grad.reshape(-1))
            enc.params.grad = None
        self._step_table(self.model.geometry.active_levels)
        self.global_step += 1
        self.last = {"loss": loss.detach(), "n_samples": n_samples,
                     "n_rays": batch["rays"].shape[0], **{k: v.detach() for k, v in terms.items()}}
        return self.last

    def fit(self, dataset, max_steps=None, log_every=0):
        self.dataset = dataset
        for _ in range(max_steps or self.config.max_steps):
            r = self.training_step()
            if log_every and self.global_step % log_every == 0:
                print(f"[nsr] step {self.global_step} loss {float(r['loss']):.4f} "
                      f"rays {r['n_rays']} samples {r['n_samples']}", flush=True)
        if self.table_opt is not None:
            self.table_opt.finalize()          # masked levels: apply their accumulated decay
        if self._native is not None:
            self._native.flush_timing()

    # ----------------------------------------------------------------- export
    @torch.no_grad()
    def export_mesh(self, front_mask=None, resolution=None, with_colors=True, face_count=None):
        """OrthoNeuSSystem.export -> model.export (neus_ortho.py:183-200, neus.py:220-238,
        geometry.py:108-117) on the device: coarse 512^3 pass, iso-surface of the smoothed coarse
        binary volume, fine box = the coarse MESH's bounding box padded by 10 %, fine pass with
        the front mask, marching cubes, vertex colours from the texture network at the surface
        normal.  `front_mask`: (H,W) uint8 tensor, already rotated as ortho.py:155-156 does.
        Returns {verts (N,3) f64, faces (M,3) i64, vert_colors (N,3) or None, level, ...}."""
        from . import mesh as M
        if self.table_opt is not None:
            self.table_opt.finalize()          # loops that call training_step() directly never ran it
        self.model.eval()
        fine, coarse = M.isosurface(self.model, front_mask, resolution, face_count)
        fine["coarse_level"] = coarse["level"]
        fine["vert_colors"] = M.vertex_colors(self.model, fine["verts"].to(self.device)) \
            if with_colors and fine["verts"].shape[0] else None
        return fine

    def export_name(self, front_cutting=True):
        """save name of neus_ortho.py:184-196 for the switches applied here (resolution, face
        count label, front cutting; remeshing / thinning / smoothing / colour back-projection are
        CPU steps outside this path and leave no suffix)."""
        iso = self.model.config.geometry.isosurface
        name = f"it{self.global_step}-{iso.method}{iso.resolution}-f50000"
        return name + ("_c" if front_cutting else "")

    def export_levels(self, resolution=None):
        """The two level volumes of the export and the fine box: (coarse, fine, vmin, vmax), the
        fine box derived from the coarse mesh as the reference does (geometry.py:111-114)."""
        m = self.export_mesh(None, resolution, with_colors=False)
        dev = m["level"].device
        r = self.model.config.radius
        vmin = torch.tensor(m.get("vmin", [-r] * 3), device=dev)
        vmax = torch.tensor(m.get("vmax", [r] * 3), device=dev)
        return m["coarse_level"], m["level"], vmin, vmax
