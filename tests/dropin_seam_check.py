"""Drop-in seam check (run as a script by tests/test_dropin_seam.py, in its own process).

The REFERENCE's unmodified instant_nsr modules (2_charactor_reconstructor/instant_nsr/models/
{neus,geometry,network_utils,texture}.py) are imported with `drawingspinup_amd.shims` first on
sys.path, so their `import tinycudann as tcnn` / `from nerfacc import ...` bind to the product's
operator surface (shims/tinycudann, shims/nerfacc -> nsr/encoding.py, nsr/render.py).  The model
they build is then run on the reference-generated fixture's inputs and must reproduce the
fixture (tests/golden/nsr_step_reference.npz, made with plain stand-ins instead of the shims).

/root/reference exists only in the build container, which has no GPU, and the product's kernels
exist only on the GPU: so here the five C-ABI wrappers the shims reach on this path
(drawingspinup_amd.ops: hashgrid_encode_fwd, ray_aabb, ray_march_single_pass,
weights_from_alpha_fwd, accumulate_fwd) are replaced by the CPU oracle — TEST INFRASTRUCTURE, in
this process only.  What this pins is the seam itself: constructor signatures, keyword sets,
argument layouts / dtypes, return structures and the parameter layout the reference's code
relies on.  The kernels behind the same wrappers are checked against the same fixture on the GPU
by tests/test_gpu_nsr_reference_step.py.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/2_charactor_reconstructor"
sys.path.insert(0, ROOT)


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    gold = np.load(os.path.join(HERE, "golden", "nsr_step_reference.npz"))
    mk = _load(os.path.join(HERE, "golden", "make_nsr_step_golden.py"), "make_nsr_step_golden")
    from oracle import hashgrid as oh, nerfacc_ref as nr
    from drawingspinup_amd import ops, shims

    # ---- CPU stand-ins for the C-ABI wrappers (oracle-backed)
    def hashgrid_encode_fwd(cfg, table_f16, x, active_levels):
        lv = oh.make_levels(cfg.n_levels, cfg.log2_hashmap_size, cfg.base_resolution, cfg.per_level_scale)
        return torch.from_numpy(oh.encode(table_f16.numpy().reshape(-1, 2), x.numpy(), lv, int(active_levels)))

    def ray_aabb(rays_o, rays_d, aabb6, jitter=None, step=0.0):
        assert jitter is None
        tmin, tmax = nr.ray_aabb_intersect(rays_o.numpy(), rays_d.numpy(), np.asarray(aabb6, np.float32))
        return torch.from_numpy(tmin), torch.from_numpy(tmax)

    def ray_march_single_pass(rays_o, rays_d, t_min, t_max, aabb6, occ_binary, res, step):
        occ = None if occ_binary is None else occ_binary.numpy().astype(bool)
        ri, ts, te, cnt = nr.ray_marching(rays_o.numpy(), rays_d.numpy(), t_min.numpy(), t_max.numpy(),
                                          np.asarray(aabb6, np.float32), occ, res, step)
        cnt = torch.from_numpy(cnt)
        off = (torch.cumsum(cnt, 0, dtype=torch.int32) - cnt).contiguous()
        return torch.from_numpy(ri), torch.from_numpy(ts), torch.from_numpy(te), off, cnt

    def weights_from_alpha_fwd(alpha, offsets, counts):
        return torch.from_numpy(nr.render_weight_from_alpha(alpha.numpy(), counts.numpy()).astype(np.float32))

    def accumulate_fwd(weights, values, offsets, counts):
        ri = np.repeat(np.arange(counts.shape[0]), counts.numpy())
        out = nr.accumulate_along_rays(weights.numpy(), None if values is None else values.numpy(), ri,
                                       counts.shape[0])
        return torch.from_numpy(out.astype(np.float32))

    for name, fn in dict(hashgrid_encode_fwd=hashgrid_encode_fwd, ray_aabb=ray_aabb,
                         ray_march_single_pass=ray_march_single_pass,
                         weights_from_alpha_fwd=weights_from_alpha_fwd,
                         accumulate_fwd=accumulate_fwd).items():
        setattr(ops, name, fn)

    # ---- the reference's unmodified modules over the shims
    shims.install()
    mk.install_stubs(native_ops=False)
    sys.path.insert(0, REF)
    import tinycudann, nerfacc                                   # noqa: E401
    assert "drawingspinup_amd/shims" in tinycudann.__file__.replace(os.sep, "/")
    assert "drawingspinup_amd/shims" in nerfacc.__file__.replace(os.sep, "/")
    from drawingspinup_amd.nsr.model import Cfg, DEFAULT_MODEL_CONFIG
    import instant_nsr.systems.utils                             # noqa: F401
    from instant_nsr import models as ref_models
    from drawingspinup_amd.nsr.encoding import Encoding
    from drawingspinup_amd.nsr.render import OccupancyGrid

    torch.manual_seed(0)
    cfg = Cfg(DEFAULT_MODEL_CONFIG)
    cfg["randomized"] = False
    model = ref_models.make("neus", cfg)
    assert type(model).__module__.startswith("instant_nsr.models")          # the reference's class
    enc = model.geometry.encoding.encoding.encoding
    assert isinstance(enc, Encoding) and isinstance(model.occupancy_grid, OccupancyGrid)
    assert enc.params.dim() == 1 and enc.params.dtype == torch.float32 and enc.n_output_dims == 20
    sd = model.state_dict()
    for k in gold.files:
        if k.startswith("sd."):
            assert k[3:] in sd, "state_dict key missing under the shims: " + k
            sd[k[3:]] = torch.from_numpy(gold[k])
    g = torch.Generator().manual_seed(int(gold["table_seed"]))
    tk = [k for k, v in sd.items() if v.numel() > 1000000 and k.endswith("params")]
    assert tk == ["geometry.encoding.encoding.encoding.params"], tk
    sd[tk[0]] = (torch.rand(sd[tk[0]].numel(), generator=g) * 2 - 1) * float(gold["table_scale"])
    model.load_state_dict(sd)
    model.occupancy_grid._binary = torch.from_numpy(mk.shell_occupancy())
    model.occupancy_grid._binary_u8 = None
    model.train()
    model.randomized = False
    model.occupancy_grid.every_n_step = lambda *a, **k: None
    model.update_step(0, int(gold["step"]))
    torch.manual_seed(5)                                         # DRAW_SEED of the fixture
    out = model.forward_(torch.from_numpy(gold["rays"]))
    bad = []
    for k, v in out.items():
        ref = gold["fwd." + k]
        got = v.detach().numpy()
        if ref.dtype.kind in "iub" or k in ("points", "intervals"):
            ok = np.array_equal(got.reshape(ref.shape), ref)
        else:
            ok = np.allclose(got.reshape(ref.shape), ref, rtol=1e-5, atol=1e-6)
        if not ok:
            bad.append(k)
        print(f"{k:22s} {'ok' if ok else 'MISMATCH'}")
    # the backward seam: autograd reaches the shim's parameters through the reference's graph
    assert not bad, bad
    print("drop-in seam: reference modules over the shims reproduce the reference fixture")


if __name__ == "__main__":
    main()
