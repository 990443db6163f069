"""Generate tests/golden/nsr_reference.npz by running the REFERENCE's own instant_nsr modules
(2_charactor_reconstructor/instant_nsr/models/{geometry,neus,network_utils,texture,utils}.py,
systems/criterions.py) on the CPU in this container.

    python tests/golden/make_nsr_golden.py          # needs /root/reference

Third-party imports that cannot be installed here are stubbed: `tinycudann.Encoding` is served
by the oracle's hash-grid restatement (oracle/hashgrid.py) — so the fixture pins everything
AROUND it that is the reference's own code: ProgressiveBandHashGrid masking / level schedule,
CompositeEncoding, VanillaMLP (weight-norm, sphere init, Softplus(100)), contract_to_unisphere,
VolumeSDF.forward's finite-difference gradient / laplacian, VolumeSDF.update_step's progressive
eps, NeuS get_alpha / occ_eval_fn arithmetic, VolumeRadiance, ranking_loss / BCE.  The hash-grid
op itself stays "parity unpinned" (see oracle/hashgrid.py).
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
from oracle import hashgrid as oh  # noqa: E402


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
        g = torch.Generator().manual_seed(7)
        n = self.lv["offsets"][self.n_levels] * 2
        self.params = torch.nn.Parameter((torch.rand(n, generator=g) * 2 - 1) * 0.2)

    def forward(self, x):
        tab = self.params.detach().half().numpy().reshape(-1, 2)
        return torch.from_numpy(oh.encode(tab, x.detach().float().numpy(), self.lv, self.n_levels))


class ContractionType(enum.Enum):
    AABB = 0
    UN_BOUNDED_TANH = 1
    UN_BOUNDED_SPHERE = 2


class _Grid(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def every_n_step(self, *a, **k):
        pass


stub("tinycudann", Encoding=_Enc, Network=None, free_temporary_memory=lambda: None)
stub("nerfacc", ContractionType=ContractionType, OccupancyGrid=_Grid, ray_marching=None,
     render_weight_from_alpha=None, accumulate_along_rays=None)
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
# torch.cuda.device(rank) context used by network_utils.py:45 -> no-op on CPU
torch.cuda.device = lambda *a, **k: __import__("contextlib").nullcontext()
_orig_zeros = torch.zeros
torch.zeros = lambda *a, **k: _orig_zeros(*a, **{kk: vv for kk, vv in k.items()
                                                if not (kk == "device" and isinstance(vv, int))})

sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
from drawingspinup_amd.nsr.model import Cfg, DEFAULT_MODEL_CONFIG  # noqa: E402  (plain config dict)
import instant_nsr.systems.utils  # noqa: E402,F401
from instant_nsr import models as ref_models  # noqa: E402
from instant_nsr.systems import criterions  # noqa: E402

torch.manual_seed(0)
cfg = Cfg(DEFAULT_MODEL_CONFIG)
cfg["grid_prune"] = False
model = ref_models.make("neus", cfg)
geo = model.geometry
out = {}
g = torch.Generator().manual_seed(1)
# non-trivial network weights (sphere init zeroes the feature columns)
with torch.no_grad():
    for p in geo.network.parameters():
        p.add_(torch.randn(p.shape, generator=g) * 0.05)
    for p in model.texture.network.parameters():
        p.add_(torch.randn(p.shape, generator=g) * 0.05)
for step in (0, 1500, 2999):
    model.train()
    model.update_step(0, step)
    pts = torch.rand(257, 3, generator=g) * 2 - 1
    pts[0] = torch.tensor([1.0, -1.0, 0.9999])
    sdf, grad, feat, lap = geo(pts, with_grad=True, with_feature=True, with_laplace=True)
    k = f"s{step}."
    out.update({k + "pts": pts.numpy(), k + "sdf": sdf.detach().numpy(), k + "grad": grad.detach().numpy(),
                k + "feature": feat.detach().numpy(), k + "laplace": lap.detach().numpy(),
                k + "eps": np.float64(geo._finite_difference_eps),
                k + "level": np.int64(geo.encoding.encoding.current_level),
                k + "cos_anneal": np.float64(model.cos_anneal_ratio),
                k + "forward_level": geo.forward_level(pts).detach().numpy()})
    dirs = torch.nn.functional.normalize(torch.randn(257, 3, generator=g), dim=-1)
    normal = torch.nn.functional.normalize(grad.detach(), dim=-1)
    dists = torch.full((257, 1), model.render_step_size)
    out[k + "dirs"] = dirs.numpy()
    out[k + "alpha"] = model.get_alpha(sdf.detach(), normal, dirs, dists).detach().numpy()
    out[k + "rgb"] = model.texture(feat.detach(), dirs, normal).detach().numpy()
out["render_step_size"] = np.float64(model.render_step_size)
# effective (weight-normed) geometry MLP weights + texture weights + table: the inputs of the oracle
lin0, lin2 = geo.network.layers[0], geo.network.layers[2]
_ = geo.network(torch.zeros(1, 23))      # refresh .weight from weight_g / weight_v
out.update({"w0": lin0.weight.detach().numpy(), "b0": lin0.bias.detach().numpy(),
            "w1": lin2.weight.detach().numpy(), "b1": lin2.bias.detach().numpy(),
            "w0_g": lin0.weight_g.detach().numpy(), "w0_v": lin0.weight_v.detach().numpy(),
            "variance": model.variance.variance.detach().numpy(),
            "table_seed": np.int64(7), "table_scale": np.float64(0.2)})
for i in (0, 2, 4):
    out[f"tex.w{i}"] = model.texture.network.layers[i].weight.detach().numpy()
    out[f"tex.b{i}"] = model.texture.network.layers[i].bias.detach().numpy()
out["state_dict_keys"] = np.array(sorted(model.state_dict().keys()))
# criterions
err = torch.rand(100, generator=g)
w = torch.rand(100, generator=g)
out.update({"rank.err": err.numpy(), "rank.w": w.numpy(),
            "rank.mean08": criterions.ranking_loss(err, 0.8, None, "mean").numpy(),
            "rank.sum09w": criterions.ranking_loss(err, 0.9, w, "sum").numpy(),
            "bce": criterions.binary_cross_entropy(err.clamp(1e-3, 1 - 1e-3), (w > 0.5).float(),
                                                   reduction="none").numpy()})
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "nsr_reference.npz"), **out)
print("wrote nsr_reference.npz with", len(out), "arrays; keys:", list(out["state_dict_keys"])[:6], "...")
