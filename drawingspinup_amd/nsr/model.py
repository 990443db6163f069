"""NeuS model on the fused gfx950 kernels.

Mirrors 2_charactor_reconstructor/instant_nsr/models/{neus,geometry,texture,network_utils}.py:
same module tree (hence the same state_dict keys: geometry.encoding.encoding.encoding.params,
geometry.network.layers.{0,2}.{weight_g,weight_v,bias}, texture.network.layers.{0,2,4}.*,
variance.variance), same forward outputs.  The per-point work of VolumeSDF.forward
(geometry.py:135-187: 7 hash-grid + MLP evaluations) is one fused kernel forward and one
backward.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .encoding import ProgressiveBandHashGrid
from .render import (ContractionType, OccupancyGrid, accumulate_along_rays, ray_marching,
                     render_weight_from_alpha)


class Cfg(dict):
    """dict with attribute access (stands in for OmegaConf nodes)."""
    __getattr__ = dict.get

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in {**(d or {}), **kw}.items():
            self[k] = Cfg(v) if isinstance(v, dict) else v


DEFAULT_MODEL_CONFIG = Cfg({   # configs/neuralangelo-ortho-wmask.yaml:22-84
    "name": "neus", "radius": 1.0, "num_samples_per_ray": 1024, "train_num_rays": 256,
    "max_train_num_rays": 8192, "grid_prune": True, "grid_prune_occ_thre": 0.001,
    "dynamic_ray_sampling": True, "batch_image_sampling": True, "randomized": True,
    "ray_chunk": 2048, "cos_anneal_end": 20000,
    "variance": {"init_val": 0.3, "modulate": False},
    "geometry": {
        "name": "volume-sdf", "radius": 1.0, "feature_dim": 13,
        "grad_type": "finite_difference", "finite_difference_eps": "progressive",
        "isosurface": {"method": "mc", "resolution": 512, "chunk": 2097152, "threshold": 0.0},
        "xyz_encoding_config": {
            "otype": "ProgressiveBandHashGrid", "n_levels": 10, "n_features_per_level": 2,
            "log2_hashmap_size": 19, "base_resolution": 32,
            "per_level_scale": 1.3195079107728942, "include_xyz": True, "start_level": 4,
            "start_step": 0, "update_steps": 1000},
        "mlp_network_config": {
            "otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none",
            "n_neurons": 64, "n_hidden_layers": 1, "sphere_init": True,
            "sphere_init_radius": 0.5, "weight_norm": True}},
    "texture": {
        "name": "volume-radiance", "input_feature_dim": 16,
        "mlp_network_config": {"otype": "VanillaMLP", "activation": "ReLU",
                               "output_activation": "none", "n_neurons": 64,
                               "n_hidden_layers": 2},
        "color_activation": "sigmoid"},
})


class VanillaMLP(nn.Module):
    """network_utils.py:94-138 (same init, same parameter names)."""

    def __init__(self, dim_in, dim_out, config):
        super().__init__()
        self.n_neurons, self.n_hidden_layers = config["n_neurons"], config["n_hidden_layers"]
        self.sphere_init = config.get("sphere_init", False)
        self.weight_norm = config.get("weight_norm", False)
        self.sphere_init_radius = config.get("sphere_init_radius", 0.5)
        layers = [self.make_linear(dim_in, self.n_neurons, True, False), self.make_activation()]
        for _ in range(self.n_hidden_layers - 1):
            layers += [self.make_linear(self.n_neurons, self.n_neurons, False, False),
                       self.make_activation()]
        layers += [self.make_linear(self.n_neurons, dim_out, False, True)]
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        return self.layers(x.float())

    def make_linear(self, dim_in, dim_out, is_first, is_last):
        layer = nn.Linear(dim_in, dim_out, bias=True)
        if self.sphere_init:
            if is_last:
                nn.init.constant_(layer.bias, -self.sphere_init_radius)
                nn.init.normal_(layer.weight, mean=math.sqrt(math.pi) / math.sqrt(dim_in),
                                std=0.0001)
            elif is_first:
                nn.init.constant_(layer.bias, 0.0)
                nn.init.constant_(layer.weight[:, 3:], 0.0)
                nn.init.normal_(layer.weight[:, :3], 0.0, math.sqrt(2) / math.sqrt(dim_out))
            else:
                nn.init.constant_(layer.bias, 0.0)
                nn.init.normal_(layer.weight, 0.0, math.sqrt(2) / math.sqrt(dim_out))
        else:
            nn.init.constant_(layer.bias, 0.0)
            nn.init.kaiming_uniform_(layer.weight, nonlinearity="relu")
        if self.weight_norm:
            layer = nn.utils.weight_norm(layer)
        return layer

    def make_activation(self):
        return nn.Softplus(beta=100) if self.sphere_init else nn.ReLU(inplace=True)

    def effective_weights(self):
        """[(W, b)] with weight-norm applied (differentiable w.r.t. weight_g / weight_v)."""
        out = []
        for m in self.layers:
            if isinstance(m, nn.Linear):
                if hasattr(m, "weight_g"):
                    w = torch._weight_norm(m.weight_v, m.weight_g, 0)
                else:
                    w = m.weight
                out.append((w, m.bias))
        return out


class CompositeEncoding(nn.Module):
    def __init__(self, encoding, include_xyz=False, xyz_scale=1.0, xyz_offset=0.0):
        super().__init__()
        self.encoding = encoding
        self.include_xyz, self.xyz_scale, self.xyz_offset = include_xyz, xyz_scale, xyz_offset
        self.n_output_dims = int(include_xyz) * encoding.n_input_dims + encoding.n_output_dims

    def forward(self, x):
        e = self.encoding(x)
        return torch.cat([x * self.xyz_scale + self.xyz_offset, e], -1) if self.include_xyz else e

    def update_step(self, epoch, global_step):
        self.encoding.update_step(epoch, global_step)


class _SdfFdFn(torch.autograd.Function):
    """Fused VolumeSDF.forward (sdf, grad, feature, laplace) with table/MLP gradients."""

    @staticmethod
    def forward(ctx, pts, params, w0, b0, w1, b1, enc, radius, eps, active, need):
        table = enc.table_f16()
        mlp = [w0.detach().contiguous(), b0.detach().contiguous(), w1.detach().contiguous(),
               b1.detach().contiguous()]
        sdf, grad, feat, lap, cache = ops.sdf_fd_fwd(enc.cfg, table, mlp, pts, radius, eps, active,
                                                     need[0], need[1], need[2], enc_cache=True)
        ctx.save_for_backward(pts, table, cache, *mlp)
        ctx.meta = (enc.cfg, radius, eps, active)
        outs = [sdf] + [t for t in (grad, feat, lap) if t is not None]
        ctx.need = need
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        pts, table, cache, w0, b0, w1, b1 = ctx.saved_tensors
        cfg, radius, eps, active = ctx.meta
        it = iter(gouts)
        d_sdf = next(it)
        d_grad = next(it) if ctx.need[0] else None
        d_feat = next(it) if ctx.need[1] else None
        d_lap = next(it) if ctx.need[2] else None
        gt, g = ops.sdf_fd_bwd(cfg, table, [w0, b0, w1, b1], pts, radius, eps, active, d_sdf,
                               d_grad, d_feat, d_lap, enc_cache=cache)
        return None, gt, g[0], g[1], g[2], g[3], None, None, None, None, None


class _SplitKLinearFn(torch.autograd.Function):
    """y = x W^T + b for tall-skinny x (N ~ 2.6e5 rows, <= 64 columns).  The weight gradient
    dW = dy^T x contracts over N with a 64x64 (or smaller) output: as ONE GEMM that is two
    workgroups of work (measured 0.3-0.6 ms in the library).  It is evaluated as a batched GEMM
    over S row-chunks (split-K) plus a sum, which fills the chip."""
    SPLIT = 256

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        n = x.shape[0]
        dx = dy @ w
        s = _SplitKLinearFn.SPLIT
        q = n // s
        if q >= 16:
            m = s * q
            dw = torch.bmm(dy[:m].view(s, q, -1).transpose(1, 2), x[:m].view(s, q, -1)).sum(0)
            if m < n:
                dw = dw + dy[m:].t() @ x[m:]
        else:
            dw = dy.t() @ x
        return dx, dw, dy.sum(0)


class _TextureFn(torch.autograd.Function):
    """sigmoid(VanillaMLP 16->64->64->3 (x)) on the fused MFMA kernels (texture.py:20-30)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, w2, b2):
        params = [t.detach().contiguous() for t in (w0, b0, w1, b1, w2, b2)]
        rgb = ops.texture_fwd(params, x)
        ctx.save_for_backward(x, rgb, *params)
        return rgb

    @staticmethod
    def backward(ctx, d_rgb):
        x, rgb, *params = ctx.saved_tensors
        d_x, g = ops.texture_bwd(params, x, rgb, d_rgb.contiguous())
        return (d_x, *g)


class _ShadePrepFn(torch.autograd.Function):
    """normal = F.normalize(sdf_grad); tex_in = cat(feature, normal)  (neus.py:143, texture.py:22)."""

    @staticmethod
    def forward(ctx, grad, feature):
        ctx.save_for_backward(grad)
        return ops.shade_prep_fwd(grad, feature)

    @staticmethod
    def backward(ctx, d_normal, d_tex_in):
        (grad,) = ctx.saved_tensors
        return ops.shade_prep_bwd(grad, d_normal.contiguous(), d_tex_in.contiguous())


class _CompositeFn(torch.autograd.Function):
    """get_alpha + render_weight_from_alpha + 4x accumulate_along_rays fused (neus.py:90-112,
    144, 147-153).  Returns comp (R,8) = [opacity, depth, rgb(3), sum w*normal(3)], weights, alpha."""

    @staticmethod
    def forward(ctx, sdf, normal, rgb, inv_s, rays_d, t_starts, t_ends, off, cnt, car):
        comp, alpha, w = ops.neus_composite_fwd(sdf, normal, rgb, rays_d, t_starts, t_ends, off,
                                                cnt, inv_s, car)
        ctx.save_for_backward(sdf, normal, rgb, inv_s, rays_d, t_starts, t_ends, off, cnt, alpha, w)
        ctx.car = car
        ctx.mark_non_differentiable(alpha)
        return comp, w, alpha

    @staticmethod
    def backward(ctx, d_comp, d_w, _d_alpha):
        sdf, normal, rgb, inv_s, rays_d, ts, te, off, cnt, alpha, w = ctx.saved_tensors
        d_sdf, d_normal, d_rgb, d_inv = ops.neus_composite_bwd(
            sdf, normal, rgb, rays_d, ts, te, off, cnt, inv_s, ctx.car, alpha, w,
            d_comp.contiguous(), None if d_w is None else d_w.contiguous())
        return d_sdf, d_normal, d_rgb, d_inv.view_as(inv_s), None, None, None, None, None, None


class VolumeSDF(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.radius = config.radius
        self.n_output_dims = config.feature_dim
        ec = config.xyz_encoding_config
        assert ec.otype == "ProgressiveBandHashGrid"
        pb = ProgressiveBandHashGrid(3, dict(ec))
        self.encoding = CompositeEncoding(pb, include_xyz=ec.get("include_xyz", False),
                                          xyz_scale=2.0, xyz_offset=-1.0)
        self.network = VanillaMLP(self.encoding.n_output_dims, self.n_output_dims,
                                  dict(config.mlp_network_config))
        assert self.network.n_hidden_layers == 1 and self.network.n_neurons == 64 \
            and self.n_output_dims == 13 and self.network.sphere_init, \
            "the fused kernel is built for the reference geometry MLP (23->64 softplus ->13)"
        self.grad_type = config.grad_type
        assert self.grad_type == "finite_difference"
        self.finite_difference_eps = config.get("finite_difference_eps", 1e-3)
        self._finite_difference_eps = None
        self.contraction_type = None

    @property
    def hashgrid(self):
        return self.encoding.encoding.encoding

    @property
    def active_levels(self):
        return self.encoding.encoding.current_level

    def _mlp(self):
        (w0, b0), (w1, b1) = self.network.effective_weights()
        return w0, b0, w1, b1

    def forward(self, points, with_grad=True, with_feature=True, with_laplace=False):
        enc = self.hashgrid
        shp = points.shape[:-1]
        pts = points.reshape(-1, 3).float().contiguous()
        w0, b0, w1, b1 = self._mlp()
        train = self.training and torch.is_grad_enabled()
        if not with_grad:
            # sdf (+feature) only: single evaluation
            n_out = 13 if with_feature else 1
            with torch.no_grad():
                out = ops.sdf_fwd(enc.cfg, enc.table_f16(),
                                  [t.detach().contiguous() for t in (w0, b0, w1, b1)], pts,
                                  self.radius, self.active_levels, n_out)
            rv = [out[:, 0].view(shp)]
            if with_feature:
                rv.append(out.view(*shp, 13))
            return rv[0] if len(rv) == 1 else rv
        need = (True, bool(with_feature), bool(with_laplace))
        eps = self._finite_difference_eps
        if train:
            outs = _SdfFdFn.apply(pts, enc.params, w0, b0, w1, b1, enc, self.radius, eps,
                                  self.active_levels, need)
        else:
            with torch.no_grad():
                o = ops.sdf_fd_fwd(enc.cfg, enc.table_f16(),
                                   [t.detach().contiguous() for t in (w0, b0, w1, b1)], pts,
                                   self.radius, eps, self.active_levels, *need)
            outs = [t for t in o if t is not None]
        it = iter(outs)
        rv = [next(it).view(shp), next(it).view(*shp, 3)]
        if with_feature:
            rv.append(next(it).view(*shp, 13))
        if with_laplace:
            rv.append(next(it).view(shp))
        return rv

    @torch.no_grad()
    def forward_level(self, points):
        enc = self.hashgrid
        w = [t.detach().contiguous() for t in self._mlp()]
        pts = points.reshape(-1, 3).float().contiguous()
        return ops.sdf_fwd(enc.cfg, enc.table_f16(), w, pts, self.radius, self.active_levels,
                           1).view(points.shape[:-1])

    def update_step(self, epoch, global_step):
        self.encoding.update_step(epoch, global_step)
        if isinstance(self.finite_difference_eps, float):
            self._finite_difference_eps = self.finite_difference_eps
        elif self.finite_difference_eps == "progressive":
            hg = self.config.xyz_encoding_config
            current_level = min(hg.start_level + max(global_step - hg.start_step, 0)
                                // hg.update_steps, hg.n_levels)
            grid_res = hg.base_resolution * hg.per_level_scale ** (current_level - 1)
            self._finite_difference_eps = 2 * self.config.radius / grid_res
        else:
            raise ValueError(f"Unknown finite_difference_eps={self.finite_difference_eps}")

    def regularizations(self, out):
        return {}


class VolumeRadiance(nn.Module):
    """texture.py:9-30 — the view direction argument is accepted and ignored, as there."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.n_output_dims = 3
        self.n_input_dims = config.input_feature_dim
        self.network = VanillaMLP(self.n_input_dims, 3, dict(config.mlp_network_config))

    def forward(self, features, dirs, *args):
        inp = torch.cat([features.view(-1, features.shape[-1])]
                        + [a.view(-1, a.shape[-1]) for a in args], -1)
        color = self.network(inp).view(*features.shape[:-1], 3).float()
        if "color_activation" in self.config:
            assert self.config.color_activation == "sigmoid"
            color = torch.sigmoid(color)
        return color

    @property
    def fused_ok(self):
        """the fused kernel is built for the reference network: 16 -> 64 -> 64 -> 3, ReLU,
        sigmoid colour activation, no weight norm"""
        n = self.network
        return (self.n_input_dims == 16 and n.n_neurons == 64 and n.n_hidden_layers == 2
                and not n.weight_norm and not n.sphere_init
                and self.config.get("color_activation") == "sigmoid")

    def fused_params(self):
        return [p for m in self.network.layers if isinstance(m, nn.Linear)
                for p in (m.weight, m.bias)]

    def rgb_fused(self, tex_in):
        """sigmoid(self.network(tex_in)) in one launch forward, two backward."""
        return _TextureFn.apply(tex_in, *self.fused_params())

    def mlp_split_k(self, x):
        """self.network(x) with the split-K weight-gradient GEMMs (same values)."""
        layers = self.network.layers
        for i, m in enumerate(layers):
            if isinstance(m, nn.Linear):
                x = _SplitKLinearFn.apply(x, m.weight, m.bias)
            else:
                x = torch.relu(x)
        return x

    def regularizations(self, out):
        return {}


class VarianceNetwork(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.init_val = config.init_val
        self.register_parameter("variance", nn.Parameter(torch.tensor(float(config.init_val))))
        assert not config.get("modulate", False)

    @property
    def inv_s(self):
        return torch.exp(self.variance * 10.0)

    def forward(self, x):
        return torch.ones([len(x), 1], device=self.variance.device) * self.inv_s


class NeuSModel(nn.Module):
    """neus.py:43-241 (NeuSModelTextureMLP)."""

    def __init__(self, config=None):
        super().__init__()
        self.config = config = Cfg(config) if config is not None else DEFAULT_MODEL_CONFIG
        self.geometry = VolumeSDF(config.geometry)
        self.texture = VolumeRadiance(config.texture)
        self.geometry.contraction_type = ContractionType.AABB
        self.variance = VarianceNetwork(config.variance)
        r = config.radius
        self.register_buffer("scene_aabb", torch.as_tensor([-r, -r, -r, r, r, r],
                                                           dtype=torch.float32))
        self._aabb_host = [-r, -r, -r, r, r, r]
        if config.grid_prune:
            self.occupancy_grid = OccupancyGrid(roi_aabb=self._aabb_host, resolution=128,
                                                contraction_type=ContractionType.AABB)
        self.randomized = config.randomized
        self.render_step_size = 1.732 * 2 * config.radius / config.num_samples_per_ray
        self.cos_anneal_ratio = 1.0
        # True: shading + compositing in the fused kernels; False: the op-by-op path through the
        # nerfacc-compatible operators (same results; kept for the drop-in seam and as a cross-check)
        self.fused_shading = True

    def occ_eval_fn(self, x):
        sdf = self.geometry(x, with_grad=False, with_feature=False)
        inv_s = self.variance(torch.zeros([1, 3]))[:, :1].clip(1e-6, 1e6)
        inv_s = inv_s.expand(sdf.shape[0], 1)
        next_sdf = sdf[..., None] - self.render_step_size * 0.5
        prev_sdf = sdf[..., None] + self.render_step_size * 0.5
        prev_cdf = torch.sigmoid(prev_sdf * inv_s)
        next_cdf = torch.sigmoid(next_sdf * inv_s)
        p, c = prev_cdf - next_cdf, prev_cdf
        return ((p + 1e-5) / (c + 1e-5)).view(-1, 1).clip(0.0, 1.0)

    def update_step(self, epoch, global_step):
        self.geometry.update_step(epoch, global_step)
        cos_anneal_end = self.config.get("cos_anneal_end", 0)
        self.cos_anneal_ratio = 1.0 if cos_anneal_end == 0 else min(1.0, global_step / cos_anneal_end)
        if self.training and self.config.grid_prune:
            self.occupancy_grid.every_n_step(step=global_step, occ_eval_fn=self.occ_eval_fn,
                                             occ_thre=self.config.get("grid_prune_occ_thre", 0.01))

    def get_alpha(self, sdf, normal, dirs, dists):
        inv_s = self.variance(torch.zeros([1, 3]))[:, :1].clip(1e-6, 1e6)
        inv_s = inv_s.expand(sdf.shape[0], 1)
        true_cos = (dirs * normal).sum(-1, keepdim=True)
        iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - self.cos_anneal_ratio)
                     + F.relu(-true_cos) * self.cos_anneal_ratio)
        next_sdf = sdf[..., None] + iter_cos * dists.reshape(-1, 1) * 0.5
        prev_sdf = sdf[..., None] - iter_cos * dists.reshape(-1, 1) * 0.5
        prev_cdf = torch.sigmoid(prev_sdf * inv_s)
        next_cdf = torch.sigmoid(next_sdf * inv_s)
        p, c = prev_cdf - next_cdf, prev_cdf
        return ((p + 1e-5) / (c + 1e-5)).view(-1).clip(0.0, 1.0)

    def forward_(self, rays, jitter=None, pts_random=None, perturb=None):
        n_rays = rays.shape[0]
        rays_o, rays_d = rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous()
        with torch.no_grad():
            ray_indices, t_starts, t_ends = ray_marching(
                rays_o, rays_d, scene_aabb=self._aabb_host,
                grid=self.occupancy_grid if self.config.grid_prune else None,
                render_step_size=self.render_step_size, stratified=self.randomized,
                cone_angle=0.0, alpha_thre=0.0, jitter=jitter)
        t_origins, t_dirs = rays_o[ray_indices], rays_d[ray_indices]
        midpoints = (t_starts + t_ends) / 2.0
        positions = t_origins + t_dirs * midpoints
        dists = t_ends - t_starts
        n_s = positions.shape[0]
        if self.training:
            # the reference evaluates geometry three times per step (samples, 2048 random
            # points, their perturbed copies: neus.py:139,155-162); same arithmetic, ONE
            # fused launch forward and one backward for all of them
            if pts_random is None:
                pts_random = torch.rand([1024 * 2, 3], device=positions.device) * 2 - 1
            if perturb is None:
                perturb = torch.randn_like(pts_random)
            n_r = pts_random.shape[0]
            allp = torch.cat([positions, pts_random, pts_random + perturb * 1e-2], 0)
            a_sdf, a_grad, a_feat, a_lap = self.geometry(allp, with_grad=True, with_feature=True,
                                                         with_laplace=True)
            sdf, sdf_grad, feature, sdf_laplace = a_sdf[:n_s], a_grad[:n_s], a_feat[:n_s], a_lap[:n_s]
            random_sdf, random_sdf_grad = a_sdf[n_s:n_s + n_r], a_grad[n_s:n_s + n_r]
            normal_perturb = a_grad[n_s + n_r:]
        else:
            sdf, sdf_grad, feature, sdf_laplace = self.geometry(positions, with_grad=True,
                                                                with_feature=True,
                                                                with_laplace=True)
        if self.fused_shading:
            from .render import RayPacking
            _, off, cnt = RayPacking.last
            normal, tex_in = _ShadePrepFn.apply(sdf_grad.contiguous(), feature.contiguous())
            if self.texture.fused_ok:
                rgb = self.texture.rgb_fused(tex_in)
            else:
                rgb = torch.sigmoid(self.texture.mlp_split_k(tex_in))
            comp, weights, alpha = _CompositeFn.apply(
                sdf.contiguous(), normal, rgb.contiguous(), self.variance.inv_s.reshape(1),
                rays_d, t_starts.reshape(-1), t_ends.reshape(-1), off, cnt,
                float(self.cos_anneal_ratio))
            opacity, depth, comp_rgb = comp[:, 0:1], comp[:, 1:2], comp[:, 2:5]
            comp_normal = F.normalize(comp[:, 5:8], p=2, dim=-1)
            weights = weights[:, None]
            comp_raw = comp
        else:
            normal = F.normalize(sdf_grad, p=2, dim=-1)
            alpha = self.get_alpha(sdf, normal, t_dirs, dists)[..., None]
            rgb = self.texture(feature, t_dirs, normal)
            weights = render_weight_from_alpha(alpha, ray_indices=ray_indices, n_rays=n_rays)
            opacity = accumulate_along_rays(weights, ray_indices, values=None, n_rays=n_rays)
            depth = accumulate_along_rays(weights, ray_indices, values=midpoints, n_rays=n_rays)
            comp_rgb = accumulate_along_rays(weights, ray_indices, values=rgb, n_rays=n_rays)
            comp_normal = accumulate_along_rays(weights, ray_indices, values=normal, n_rays=n_rays)
            comp_normal = F.normalize(comp_normal, p=2, dim=-1)
        out = {} if not (self.fused_shading and self.training) else {"comp_raw": comp_raw}
        out.update({"comp_rgb": comp_rgb, "comp_normal": comp_normal, "opacity": opacity,
               "depth": depth, "rays_valid": opacity > 0,
               "num_samples": torch.as_tensor([len(t_starts)], dtype=torch.int32,
                                              device=rays.device)})
        if self.training:
            out.update({"sdf_samples": sdf, "sdf_grad_samples": sdf_grad,
                        "random_sdf": random_sdf, "random_sdf_grad": random_sdf_grad,
                        "normal_perturb": normal_perturb, "weights": weights.view(-1),
                        "points": midpoints.view(-1), "intervals": dists.view(-1),
                        "ray_indices": ray_indices.view(-1),
                        "sdf_laplace_samples": sdf_laplace})
        return out

    def forward(self, rays, **kw):
        if self.training:
            out = self.forward_(rays, **kw)
        else:
            outs = [self.forward_(rays[i:i + self.config.ray_chunk])
                    for i in range(0, rays.shape[0], self.config.ray_chunk)]
            out = {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}
        return {**out, "inv_s": self.variance.inv_s}

    def train(self, mode=True):
        self.randomized = mode and self.config.randomized
        return super().train(mode=mode)

    def eval(self):
        self.randomized = False
        return super().eval()

    def regularizations(self, out):
        return {}

    @torch.no_grad()
    def isosurface_levels(self, vmin, vmax, resolution=None, chunk=None, lattice_kernel=None):
        """BaseImplicitGeometry.isosurface_ grid evaluation (geometry.py:83-106): SDF on the
        res^3 lattice, x-major / y / z-minor ('ij' meshgrid order, geometry.py:44-46),
        evaluated chunk by chunk on the device.  Returns (res,res,res) f32 on the device."""
        iso = self.config.geometry.isosurface
        res = resolution or iso.resolution
        if lattice_kernel is None:
            lattice_kernel = chunk is None       # an explicit chunk size asks for the chunked tensor expression
        chunk = chunk or iso.chunk
        dev = self.scene_aabb.device
        lin = torch.linspace(0, 1, res, device=dev)
        level = torch.empty(res ** 3, dtype=torch.float32, device=dev)
        vmin = [float(v) for v in vmin]
        vmax = [float(v) for v in vmax]
        # a chunk = `chunk // res^2` x-slabs (chunk is a multiple of res^2 for 512/2097152)
        if dev.type == "cuda" and lattice_kernel:
            # the lattice points are formed in the kernel (dsu_sdf_fwd_lattice: lin[i] * span + lo per
            # axis, the two rounded float32 operations of the tensor expression below)
            lo = torch.tensor(vmin, dtype=torch.float64).float()
            span = (torch.tensor(vmax, dtype=torch.float64) - torch.tensor(vmin, dtype=torch.float64)).float()
            g = self.geometry
            w = [t.detach().contiguous() for t in g._mlp()]
            ops.sdf_fwd_lattice(g.hashgrid.cfg, g.hashgrid.table_f16(), w, lin.contiguous(), 0, res,
                                lo.tolist(), span.tolist(), g.radius, g.active_levels, level)
            return level.view(res, res, res)
        slab = max(1, chunk // (res * res))
        yz = torch.stack(torch.meshgrid(lin, lin, indexing="ij"), -1).reshape(-1, 2)
        for x0 in range(0, res, slab):
            xs = lin[x0:x0 + slab]
            pts = torch.cat([xs[:, None, None].expand(-1, yz.shape[0], 1),
                             yz[None].expand(xs.shape[0], -1, -1)], -1).reshape(-1, 3)
            # scale_anything(x, (0,1), (vmin, vmax)) with the reference's type promotion
            # (geometry.py:85-89): the box corners are float64 scalars, so the span is formed
            # in float64 and only then rounded to the float32 the lattice is multiplied in
            lo = torch.tensor(vmin, dtype=torch.float64).float().to(dev)
            span = (torch.tensor(vmax, dtype=torch.float64)
                    - torch.tensor(vmin, dtype=torch.float64)).float().to(dev)
            pts = (pts - 0.0) / (1.0 - 0.0) * span + lo
            level[x0 * res * res:(x0 + xs.shape[0]) * res * res] = self.geometry.forward_level(pts)
        return level.view(res, res, res)
