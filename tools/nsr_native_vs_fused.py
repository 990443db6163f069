"""Sanity run over the whole level schedule (3000 steps: 4 -> 5 -> 6 levels, ~187 occupancy
refreshes) on the synthetic sphere, whose ground truth is analytic: the natively sequenced step
against the Python-sequenced step.  Their random draws differ (Philox vs torch), so the comparison
is statistical: loss terms over the last steps, and direct quality measures against the sphere —
|sdf| and normal error at ground-truth surface points, colour error there, radius of the exported
mesh.

modes:  native         dsu_nsr_driver_step, own draws, prefetch
        fused          Python-sequenced step, torch draws
        native-inject  native step fed torch draws (no prefetch)   -> isolates the draw stream
        fused-philox   Python-sequenced step fed dsu_nsr_draws     -> isolates the sequencing
        autograd       the op-by-op step through torch.autograd (the path the oracle / reference
                       fixtures pin operator by operator), torch draws
        fused-torch    Python-sequenced fused step fed the SAME torch draws as `autograd`
usage: nsr_native_vs_fused.py [steps] [mode,mode,...] [seed]"""
import ctypes as C
import json, os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from drawingspinup_amd import _lib, ops
from drawingspinup_amd.nsr import mesh as M
from drawingspinup_amd.nsr.system import OrthoData, OrthoNeuSSystem

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
order = sys.argv[2].split(",") if len(sys.argv) > 2 else ["native", "fused"]
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")


def torch_draws(ds, n, g):
    return {"index": torch.randint(0, len(ds.all_masks), (n,), generator=g, device=dev),
            "x": torch.randint(0, ds.w, (n,), generator=g, device=dev),
            "y": torch.randint(0, ds.h, (n,), generator=g, device=dev),
            "jitter": torch.rand(n, generator=g, device=dev),
            "pts_random": torch.rand(2048, 3, generator=g, device=dev) * 2 - 1,
            "perturb": torch.randn(2048, 3, generator=g, device=dev)}


def philox_draws(ds, n, step):
    idx, x, y = (torch.empty(n, dtype=torch.int64, device=dev) for _ in range(3))
    jit = torch.empty(n, device=dev)
    pr, pe = torch.empty(2048, 3, device=dev), torch.empty(2048, 3, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    ops.check(_lib.lib().dsu_nsr_draws(seed, step, n, len(ds.all_masks), ds.h, ds.w, p(idx), p(x),
                                       p(y), p(jit), 2048, p(pr), p(pe), ops.stream()), "draws")
    return {"index": idx, "x": x, "y": y, "jitter": jit, "pts_random": pr, "perturb": pe}


def fibonacci_sphere(n, r):
    i = torch.arange(n, dtype=torch.float64) + 0.5
    phi = torch.acos(1 - 2 * i / n)
    th = torch.pi * (1 + 5 ** 0.5) * i
    return (r * torch.stack([torch.cos(th) * torch.sin(phi), torch.sin(th) * torch.sin(phi),
                             torch.cos(phi)], -1)).float().to(dev)


def quality(sysm):
    m = sysm.model
    m.eval()
    with torch.no_grad():
        p = fibonacci_sphere(20000, 0.5)
        sdf, grad, feat = m.geometry(p, with_grad=True, with_feature=True)
        n = torch.nn.functional.normalize(grad, dim=-1)
        cosang = (n * p / 0.5).sum(-1).clamp(-1, 1)
        col = M.vertex_colors(m, p)
        want = 0.5 + 0.4 * p / 0.5
        out = {"sdf_mean": float(sdf.mean()), "sdf_abs": float(sdf.abs().mean()),
               "normal_deg": float(torch.rad2deg(torch.acos(cosang)).mean()),
               "grad_norm": float(grad.norm(dim=-1).mean()),
               "colour_rmse": float((col - want).pow(2).mean().sqrt()),
               "inv_s": float(m.variance.inv_s)}
    mesh = sysm.export_mesh(None, 256, with_colors=False)
    rad = mesh["verts"].float().norm(dim=1)
    out.update({"verts": int(rad.numel()), "radius_mean": float(rad.mean()), "radius_std": float(rad.std())})
    return out


out = {}
for mode in ["warm-native", "warm-fused"] + order:
    warm = mode.startswith("warm-")
    name = mode[5:] if warm else mode
    ds = OrthoData.synthetic_sphere(512, device=dev)
    sysm = OrthoNeuSSystem(device=dev, seed=seed)
    sysm.dataset = ds
    sysm.step_mode = "native" if name.startswith("native") else ("autograd" if name == "autograd" else "fused")
    g = torch.Generator(device=dev).manual_seed(seed)
    torch.manual_seed(seed)
    n = 40 if warm else steps
    torch.cuda.synchronize(); t0 = time.time()
    tail, samples = [], 0
    for s in range(n):
        if name == "native-inject":
            r = sysm.training_step_native(torch_draws(ds, int(sysm.train_num_rays), g))
        elif name == "fused-philox":
            r = sysm.training_step_fused(philox_draws(ds, int(sysm.train_num_rays), s))
        elif name == "autograd":
            r = sysm.training_step_autograd(torch_draws(ds, int(sysm.train_num_rays), g))
        elif name == "fused-torch":
            r = sysm.training_step_fused(torch_draws(ds, int(sysm.train_num_rays), g))
        else:
            r = sysm.training_step()
        samples += r["n_samples"]
        if s >= n - 50:
            tail.append({k: float(r[k]) for k in ("loss", "rgb_mse", "normal", "mask", "eikonal")})
    torch.cuda.synchronize(); dt = time.time() - t0
    if warm:
        continue
    if sysm.table_opt is not None:
        sysm.table_opt.finalize()
    mean = {k: sum(t[k] for t in tail) / len(tail) for k in tail[0]}
    out[mode] = {"ms_per_step": dt / steps * 1e3, "tail50": mean, "n_rays": r["n_rays"],
                 "mean_samples": samples / steps, "levels": int(sysm.model.geometry.active_levels),
                 **quality(sysm)}
    print(mode, json.dumps(out[mode]), flush=True)
print(json.dumps(out))
