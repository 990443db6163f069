"""Generate tests/golden/nsr_step_reference.npz by running the REFERENCE's own
NeuSModelTextureMLP.forward_ (2_charactor_reconstructor/instant_nsr/models/neus.py:114-194) and
OrthoNeuSSystem.training_step (instant_nsr/systems/neus_ortho.py:79-169) on the CPU here.

    python tests/golden/make_nsr_step_golden.py        # needs /root/reference (~1 min)

Third-party imports are stubbed as in make_nsr_golden.py; in addition `nerfacc` is served by the
oracle restatement (oracle/nerfacc_ref.py: ray_marching / render_weight_from_alpha /
accumulate_along_rays) and `tinycudann.Encoding` by oracle/hashgrid.py, so the fixture pins
everything that is the reference's OWN code on this path — sample positions from
(ray_indices, t), the three geometry calls, normal / alpha / texture, the four composites and
their normalisation, the random regulariser points, and the whole loss section of training_step
(masking, cosine clamp, the three ranking losses incl. geo-aware weighting, eikonal, BCE with
clamped opacity, sparsity, 3-D normal smoothness, the weighted total).  The two third-party ops
stay "parity unpinned".  Tie order in ranking_loss: see the comment at the training_step call.

The hash table comes from a seed (as in make_nsr_golden.py); every other parameter is stored.
"""
import enum
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/2_charactor_reconstructor"
from oracle import hashgrid as oh, nerfacc_ref as nr  # noqa: E402

TABLE_SEED, TABLE_SCALE = 7, 0.2


def stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Enc(torch.nn.Module):
    """tinycudann.Encoding stand-in built on the oracle (CPU, returns half like tcnn)."""

    def __init__(self, n_input_dims, cfg):
        super().__init__()
        self.lv = oh.make_levels(cfg["n_levels"], cfg["log2_hashmap_size"], cfg["base_resolution"],
                                 cfg["per_level_scale"])
        self.n_levels = cfg["n_levels"]
        self.n_output_dims = 2 * self.n_levels
        g = torch.Generator().manual_seed(TABLE_SEED)
        n = self.lv["offsets"][self.n_levels] * 2
        self.params = torch.nn.Parameter((torch.rand(n, generator=g) * 2 - 1) * TABLE_SCALE)

    def forward(self, x):
        tab = self.params.detach().half().numpy().reshape(-1, 2)
        return torch.from_numpy(oh.encode(tab, x.detach().float().numpy(), self.lv, self.n_levels))


class ContractionType(enum.Enum):
    AABB = 0
    UN_BOUNDED_TANH = 1
    UN_BOUNDED_SPHERE = 2


class OccupancyGrid(torch.nn.Module):
    def __init__(self, roi_aabb, resolution=128, contraction_type=None):
        super().__init__()
        self.roi_aabb, self.res = roi_aabb, resolution
        self.binary = torch.zeros(resolution, resolution, resolution, dtype=torch.bool)

    def every_n_step(self, *a, **k):
        pass


def ray_marching(rays_o, rays_d, scene_aabb=None, grid=None, alpha_fn=None, near_plane=None,
                 far_plane=None, render_step_size=1e-3, stratified=False, cone_angle=0.0,
                 alpha_thre=0.0):
    assert not stratified and alpha_fn is None and cone_angle == 0.0
    o, d = rays_o.numpy(), rays_d.numpy()
    aabb = scene_aabb.numpy()
    tmin, tmax = nr.ray_aabb_intersect(o, d, aabb)
    ri, ts, te, _ = nr.ray_marching(o, d, tmin, tmax, aabb,
                                    None if grid is None else grid.binary.numpy(),
                                    128 if grid is None else grid.res, render_step_size)
    return torch.from_numpy(ri), torch.from_numpy(ts)[:, None], torch.from_numpy(te)[:, None]


def _counts(ray_indices, n_rays):
    return np.bincount(ray_indices.numpy().reshape(-1), minlength=n_rays).astype(np.int32)


def render_weight_from_alpha(alpha, ray_indices=None, n_rays=None):
    w = nr.render_weight_from_alpha(alpha.detach().numpy().reshape(-1), _counts(ray_indices, n_rays))
    return torch.from_numpy(w.astype(np.float32))[:, None]


def accumulate_along_rays(weights, ray_indices, values=None, n_rays=None):
    v = None if values is None else values.detach().numpy()
    out = nr.accumulate_along_rays(weights.detach().numpy().reshape(-1), v,
                                   ray_indices.numpy().reshape(-1), n_rays)
    return torch.from_numpy(out.astype(np.float32))


def install_stubs(native_ops=True):
    """native_ops=False leaves `tinycudann` / `nerfacc` to whatever is importable (the drop-in
    shims in tests/dropin_seam_check.py) and stubs only the host-side packages."""
    if native_ops:
        stub("tinycudann", Encoding=_Enc, Network=None, free_temporary_memory=lambda: None)
        stub("nerfacc", ContractionType=ContractionType, OccupancyGrid=OccupancyGrid,
             ray_marching=ray_marching, render_weight_from_alpha=render_weight_from_alpha,
             accumulate_along_rays=accumulate_along_rays)
    stub("pytorch_lightning", LightningModule=torch.nn.Module, LightningDataModule=object)
    stub("pytorch_lightning.utilities")
    stub("pytorch_lightning.utilities.rank_zero", rank_zero_info=lambda *a, **k: None,
         rank_zero_debug=lambda *a, **k: None, _get_rank=lambda: 0)
    oc = stub("omegaconf")

    class OmegaConf:
        @staticmethod
        def to_container(c, resolve=True):
            return {k: (OmegaConf.to_container(v) if isinstance(v, dict) else v) for k, v in c.items()}
    oc.OmegaConf = OmegaConf
    for name in ("mcubes", "cv2", "trimesh", "sklearn", "sklearn.neighbors"):
        stub(name, NearestNeighbors=None)
    stub("instant_nsr.utils.mesh_utils", remesh=None, save_mesh=None)
    torch.cuda.device = lambda *a, **k: __import__("contextlib").nullcontext()
    _orig_zeros = torch.zeros
    torch.zeros = lambda *a, **k: _orig_zeros(*a, **{kk: vv for kk, vv in k.items()
                                                    if not (kk == "device" and isinstance(vv, int))})


def shell_occupancy(res=128, r_in=0.25, r_out=0.62):
    """binary grid of the fixture: cells whose centre lies in a spherical shell (rays cross
    empty -> occupied -> empty -> occupied -> empty)."""
    c = (np.arange(res) + 0.5) / res * 2 - 1
    x, y, z = np.meshgrid(c, c, c, indexing="ij")
    rr = np.sqrt(x * x + y * y + z * z)
    return (rr >= r_in) & (rr <= r_out)


def fixture_rays(n=192, seed=3):
    """orthographic bundles from the six axis directions plus a few oblique ones."""
    g = torch.Generator().manual_seed(seed)
    uv = torch.rand(n, 2, generator=g) * 1.5 - 0.75
    axes = torch.tensor([[0, 0, -1.0], [0, 0, 1.0], [-1.0, 0, 0], [1.0, 0, 0], [0, -1.0, 0],
                         [0.6, 0.0, -0.8]])
    d = axes[torch.arange(n) % 6]
    d = torch.nn.functional.normalize(d, dim=-1)
    # an orthonormal frame per direction; origin = 1.3 behind the centre, offset in the frame
    helper = torch.where((d[:, 2:3].abs() < 0.9), torch.tensor([[0.0, 0.0, 1.0]]), torch.tensor([[1.0, 0.0, 0.0]]))
    u = torch.nn.functional.normalize(torch.cross(d, helper.expand_as(d), dim=-1), dim=-1)
    v = torch.cross(d, u, dim=-1)
    o = -1.3 * d + uv[:, :1] * u + uv[:, 1:] * v
    return torch.cat([o, d], -1)


if __name__ == "__main__":
    install_stubs()
    sys.path.insert(0, REF)
    from drawingspinup_amd.nsr.model import Cfg, DEFAULT_MODEL_CONFIG
    from drawingspinup_amd.nsr.system import DEFAULT_SYSTEM_CONFIG
    import instant_nsr.systems.utils  # noqa: F401
    from instant_nsr import models as ref_models
    from instant_nsr import systems as ref_systems
    import instant_nsr.systems.neus_ortho  # noqa: F401  (registers ortho-neus-system)

    torch.manual_seed(0)
    cfg = Cfg(DEFAULT_MODEL_CONFIG)
    cfg["randomized"] = False            # stratified jitter off: ray_marching is deterministic
    model = ref_models.make("neus", cfg)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in model.geometry.network.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.05)
        for p in model.texture.network.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.05)
    model.occupancy_grid.binary = torch.from_numpy(shell_occupancy())
    STEP = 1500
    model.train()
    model.randomized = False
    model.update_step(0, STEP)
    rays = fixture_rays()
    DRAW_SEED = 5
    torch.manual_seed(DRAW_SEED)
    fwd = model.forward_(rays)
    torch.manual_seed(DRAW_SEED)                       # the two draws inside forward_ (neus.py:153,158)
    pts_random = torch.rand([1024 * 2, 3]) * 2 - 1
    perturb = torch.randn_like(pts_random)
    out = {"step": np.int64(STEP), "rays": rays.numpy(), "pts_random": pts_random.numpy(),
           "perturb": perturb.numpy(), "table_seed": np.int64(TABLE_SEED),
           "table_scale": np.float64(TABLE_SCALE),
           "cos_anneal_ratio": np.float64(model.cos_anneal_ratio),
           "eps": np.float64(model.geometry._finite_difference_eps),
           "level": np.int64(model.geometry.encoding.encoding.current_level)}
    for k, v in fwd.items():
        out["fwd." + k] = v.detach().numpy()
    for k, v in model.state_dict().items():
        if "encoding.params" in k or k.endswith("encoding.encoding.encoding.params"):
            continue
        if v.numel() > 100000:
            continue
        out["sd." + k] = v.numpy()

    # ---- training_step on the forward_ output + a synthetic ray batch
    System = ref_systems.systems["ortho-neus-system"] if hasattr(ref_systems, "systems") else None
    if System is None:
        from instant_nsr.systems.neus_ortho import OrthoNeuSSystem as System
    R = rays.shape[0]
    gb = torch.Generator().manual_seed(9)
    u = lambda *s: torch.rand(*s, generator=gb)
    nrm = torch.nn.functional.normalize(-rays[:, 3:6] + 0.6 * (u(R, 3) * 2 - 1), dim=-1)
    nrm[::7] = torch.nn.functional.normalize(u(R // 7 + 1, 3)[: len(nrm[::7])] * 2 - 1, dim=-1)  # some back-facing
    batch = {"rays": rays, "rgb": u(R, 3), "normal": nrm, "mask": (u(R) > 0.3).float(),
             "view_weights": u(R) + 0.5}
    batch["cosines"] = torch.nn.CosineSimilarity(dim=-1, eps=1e-6)(rays[:, 3:6], nrm)
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
    leaf = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v)
            for k, v in fwd.items()}
    leaf["inv_s"] = model.variance.inv_s.detach() if hasattr(model.variance, "inv_s") \
        else model.variance(torch.zeros([1, 3]))[0, 0].detach()
    sysm.forward = lambda b: leaf
    sysm.model = model
    tb = {k: v.clone() for k, v in batch.items()}
    # criterions.py:16 sorts with torch.sort(error) (stable=False): when a run of equal errors
    # straddles the selection boundary (here: the rays that miss the shell all have opacity 0) the
    # value of ranking_loss depends on the backend's tie order.  The fixture pins the order a
    # stable sort gives (ties keep their original order), which is what the HIP kernel implements.
    _sort = torch.sort
    torch.sort = lambda x, *a, **k: _sort(x, *a, **{**k, "stable": True})
    res = sysm.training_step(tb, 0)
    torch.sort = _sort
    res["loss"].backward()
    for k, v in batch.items():
        out["batch." + k] = v.numpy()
    out["loss.total"] = res["loss"].detach().numpy()
    for k, v in logged.items():
        if k.startswith("train/loss"):
            out["loss." + k.split("/", 1)[1]] = np.asarray(float(v))
    out["train_num_rays_after"] = np.int64(sysm.train_num_rays)
    for k in ("comp_rgb", "comp_normal", "opacity", "sdf_grad_samples", "random_sdf",
              "random_sdf_grad", "normal_perturb"):
        gk = leaf[k].grad
        out["dloss." + k] = (torch.zeros_like(leaf[k]) if gk is None else gk).numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "nsr_step_reference.npz"), **out)
    print("wrote nsr_step_reference.npz:", {k: v.shape for k, v in out.items() if not k.startswith("sd.")})
    print({k: float(v) for k, v in out.items() if k.startswith("loss.")})
