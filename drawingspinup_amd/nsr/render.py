"""nerfacc==0.3.3-compatible operator surface on the gfx950 kernels.

Same names / argument meaning as the reference's imports
(2_charactor_reconstructor/instant_nsr/models/neus.py:4): ContractionType, OccupancyGrid,
ray_marching, render_weight_from_alpha, accumulate_along_rays.  `ray_marching` additionally
stashes the per-ray (offsets, counts) packing so that the compositing kernels do not have to
rebuild it from ray_indices.
"""
import enum

import torch
import torch.nn as nn

from .. import ops


class ContractionType(enum.Enum):
    AABB = 0
    UN_BOUNDED_TANH = 1
    UN_BOUNDED_SPHERE = 2


class OccupancyGrid(nn.Module):
    NUM_DIM = 3

    def __init__(self, roi_aabb, resolution=128, contraction_type=ContractionType.AABB):
        super().__init__()
        assert contraction_type == ContractionType.AABB, "only AABB grids are on the hot path"
        if isinstance(resolution, int):
            resolution = [resolution] * 3
        assert resolution[0] == resolution[1] == resolution[2]
        self.res = int(resolution[0])
        self.num_cells = self.res ** 3
        self.contraction_type = contraction_type
        self.register_buffer("_roi_aabb", torch.as_tensor(roi_aabb, dtype=torch.float32))
        self.register_buffer("resolution", torch.tensor(resolution, dtype=torch.int32))
        self.register_buffer("occs", torch.zeros(self.num_cells))
        self.register_buffer("_binary", torch.zeros(resolution, dtype=torch.bool))
        self._binary_u8 = None
        self._aabb_host = [float(v) for v in torch.as_tensor(roi_aabb).tolist()]
        # set by the native step driver: refresh(grid, step, all_cells, occ_thre, ema_decay) -> bool,
        # the whole of _update as one stream-ordered library call (dsu_nsr_driver_occ_refresh)
        self.native_refresh = None

    @property
    def roi_aabb(self):
        return self._roi_aabb

    @property
    def binary(self):
        return self._binary

    def binary_u8(self):
        if self._binary_u8 is None or self._binary_u8.device != self.occs.device:
            self._binary_u8 = self._binary.reshape(-1).to(torch.uint8).contiguous()
        return self._binary_u8

    @torch.no_grad()
    def _cell_points(self, indices, rand=None):
        """(grid_coords + U[0,1)) / res mapped back to the roi (nerfacc grid.py _update)."""
        res = self.res
        if indices is None:
            indices = torch.arange(self.num_cells, device=self.occs.device)
        ix = torch.div(indices, res * res, rounding_mode="floor")
        iy = torch.div(indices, res, rounding_mode="floor") % res
        iz = indices % res
        coords = torch.stack([ix, iy, iz], -1).float()
        if rand is None:
            rand = torch.rand_like(coords)
        x = (coords + rand) / res
        lo, hi = self._roi_aabb[:3], self._roi_aabb[3:]
        return x * (hi - lo) + lo

    @torch.no_grad()
    def _update(self, step, occ_eval_fn, occ_thre=0.01, ema_decay=0.95, warmup_steps=256,
                rand=None, indices=None):
        if (self.native_refresh is not None and rand is None and indices is None
                and self.native_refresh(self, step, step < warmup_steps, occ_thre, ema_decay)):
            return
        if indices is None and step >= warmup_steps:
            n = self.num_cells // 4
            uniform = torch.randint(self.num_cells, (n,), device=self.occs.device)
            occupied = torch.nonzero(self._binary.flatten())[:, 0]
            if n < len(occupied):
                sel = torch.randint(len(occupied), (n,), device=self.occs.device)
                occupied = occupied[sel]
            indices = torch.cat([uniform, occupied], 0)
        x = self._cell_points(indices, rand)
        occ = occ_eval_fn(x).reshape(-1)
        ops.occgrid_ema(self.occs, indices, occ, ema_decay)
        thre = torch.clamp(self.occs.mean(), max=occ_thre)
        self._binary_u8 = ops.occgrid_binarize(self.occs, float(thre))
        self._binary = self._binary_u8.view(self.res, self.res, self.res).bool()

    @torch.no_grad()
    def every_n_step(self, step, occ_eval_fn, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256,
                     n=16):
        if not self.training:
            raise RuntimeError("OccupancyGrid.every_n_step() is a training-only call")
        if step % n == 0:
            self._update(step, occ_eval_fn, occ_thre, ema_decay, warmup_steps)


class RayPacking:
    """Side channel between ray_marching and the compositing ops."""
    last = None
    total = 0      # number of samples of the last ray_marching call (host int)


@torch.no_grad()
def ray_marching(rays_o, rays_d, t_min=None, t_max=None, scene_aabb=None, grid=None,
                 sigma_fn=None, alpha_fn=None, early_stop_eps=1e-4, alpha_thre=0.0,
                 near_plane=None, far_plane=None, render_step_size=1e-3, stratified=False,
                 cone_angle=0.0, jitter=None):
    """Returns (ray_indices, t_starts (n,1), t_ends (n,1)) like nerfacc 0.3.3."""
    assert cone_angle == 0.0 and sigma_fn is None and alpha_fn is None
    aabb = [float(v) for v in scene_aabb.tolist()] if torch.is_tensor(scene_aabb) else scene_aabb
    if stratified and jitter is None:
        jitter = torch.rand(rays_o.shape[0], device=rays_o.device)
    if not stratified:
        jitter = None
    tmin, tmax = ops.ray_aabb(rays_o, rays_d, aabb, jitter, render_step_size)
    if near_plane is not None:
        tmin = torch.clamp(tmin, min=near_plane)
    if far_plane is not None:
        tmax = torch.clamp(tmax, max=far_plane)
    occ, res, gaabb = None, 0, aabb
    if grid is not None:
        occ, res, gaabb = grid.binary_u8(), grid.res, grid._aabb_host
    ri, ts, te, off, cnt = ops.ray_march_single_pass(rays_o, rays_d, tmin, tmax, gaabb, occ, res,
                                                     render_step_size)
    RayPacking.last = (ri, off, cnt)
    RayPacking.total = int(ri.shape[0])
    return ri, ts[:, None], te[:, None]


def _packing(ray_indices, n_rays):
    last = RayPacking.last
    if last is not None and last[0].data_ptr() == ray_indices.data_ptr() \
            and last[1].shape[0] == n_rays:
        return last[1], last[2]
    cnt = torch.bincount(ray_indices, minlength=n_rays).to(torch.int32)
    off = (torch.cumsum(cnt, 0, dtype=torch.int32) - cnt).contiguous()
    return off, cnt


class _WeightsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha, off, cnt):
        w = ops.weights_from_alpha_fwd(alpha, off, cnt)
        ctx.save_for_backward(alpha, w, off, cnt)
        return w

    @staticmethod
    def backward(ctx, gw):
        alpha, w, off, cnt = ctx.saved_tensors
        return ops.weights_from_alpha_bwd(alpha, w, gw.contiguous(), off, cnt), None, None


def render_weight_from_alpha(alpha, packed_info=None, ray_indices=None, n_rays=None):
    shp = alpha.shape
    off, cnt = _packing(ray_indices, n_rays)
    return _WeightsFn.apply(alpha.reshape(-1).float().contiguous(), off, cnt).view(shp)


class _AccumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, values, ray_indices, off, cnt):
        ctx.save_for_backward(w, values, ray_indices)
        return ops.accumulate_fwd(w, values, off, cnt)

    @staticmethod
    def backward(ctx, gout):
        w, values, ri = ctx.saved_tensors
        g = gout[ri]                       # (n, C)
        if values is None:
            return g[:, 0], None, None, None, None
        return (g * values).sum(-1), g * w[:, None], None, None, None


def accumulate_along_rays(weights, ray_indices, values=None, n_rays=None):
    off, cnt = _packing(ray_indices, n_rays)
    w = weights.reshape(-1).float().contiguous()
    v = None if values is None else values.float().contiguous()
    return _AccumFn.apply(w, v, ray_indices, off, cnt)
