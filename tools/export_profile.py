"""Where the export + save_mesh time of one drawing goes (bench.py: nsr_export + nsr_post).
    python tools/export_profile.py [fit_steps]
Runs a short NSR fit on the synthetic sphere data, then the export the way DrawingPipeline.reconstruct
does, with a synchronise + wall-clock around every part."""
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from drawingspinup_amd.nsr import mesh as M  # noqa: E402
from drawingspinup_amd.nsr.system import OrthoData, OrthoNeuSSystem  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ds = OrthoData.synthetic_sphere(1024, device=dev)
sysm = OrthoNeuSSystem(device=dev, seed=0)
sysm.fit(ds, max_steps=steps)
torch.cuda.synchronize()
T = {}


def timed(name, fn, *a, **k):
    torch.cuda.synchronize()
    t0 = time.time()
    out = fn(*a, **k)
    torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + time.time() - t0
    return out


for rep in range(2):
    T.clear()
    model = sysm.model
    model.eval()
    res, r = 512, 1.0
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 1024, device=dev), torch.linspace(-1, 1, 1024, device=dev), indexing="ij")
    front = (((xx / 0.55) ** 2 + (yy / 0.8) ** 2 <= 1.0) * 255).to(torch.uint8)
    front = torch.rot90(front, k=-1, dims=(0, 1)).contiguous()
    t_all = time.time()
    lvl = timed("levels_coarse", model.isosurface_levels, (-r,) * 3, (r,) * 3, res)
    binary = lvl <= 0
    val = timed("smooth_coarse", M.smooth_constrained, binary)
    v, f = timed("mc_coarse", M.marching_cubes, val, 0.0)
    v = v / (res - 1.0) * 2 - 1
    vmin, vmax = v.amin(0), v.amax(0)
    vmin_ = (vmin - (vmax - vmin) * 0.1).clamp(-r, r).tolist()
    vmax_ = (vmax + (vmax - vmin) * 0.1).clamp(-r, r).tolist()
    fm = M.crop_front_mask(front, vmin_, vmax_)
    lvl = timed("levels_fine", model.isosurface_levels, vmin_, vmax_, res)
    fmr = timed("mask_resize", M.resize_cubic_u8, fm, (res, res))
    binary = (lvl <= 0) & (fmr[:, None, :].expand(res, res, res) > 127)
    val = timed("smooth_fine", M.smooth_constrained, binary)
    v, f = timed("mc_fine", M.marching_cubes, val, 0.0)
    print("fine mesh", tuple(v.shape), tuple(f.shape))
    v = v / (res - 1.0)
    hv, hf = timed("remesh_50000", M.remesh, v, f, 50000)
    print("remesh stats", M.last_remesh_stats)
    verts = torch.from_numpy(hv).to(dev)
    faces = torch.from_numpy(hf).to(dev)
    verts = torch.stack([M.scale_anything(verts[:, a], (0, 1), (vmin_[a], vmax_[a])) for a in range(3)], -1)
    colors = torch.rand(6, 3, 256, 256, device=dev)
    big = lambda t: (F.interpolate(t[None].float(), size=(2048, 2048), mode="bicubic", align_corners=False)[0]
                     .clamp(0, 1) * 255).to(torch.uint8)
    cbp = timed("cbp_inputs", lambda: {"color_front": big(colors[0]).permute(1, 2, 0).contiguous(),
                                       "color_back": big(colors[3]).permute(1, 2, 0).contiguous(),
                                       "mask_front": big((front.float() / 255)[None])[0].contiguous()})
    # post_process_mesh, part by part
    vv = verts.detach().cpu().numpy().astype(np.float64) * 0.5
    old = np.zeros_like(vv)
    old[:, 0], old[:, 1], old[:, 2] = vv[:, 0], vv[:, 2], -vv[:, 1]
    fz = faces.cpu().numpy().astype(np.int64)
    out = timed("laplacian_smooth", M.laplacian_smooth_implicit, old, fz, lamb=2.0, iterations=5)
    from drawingspinup_amd.nsr.mesh_post import color_projection
    c = timed("color_projection", lambda: color_projection(
        torch.from_numpy(np.ascontiguousarray(out)).to(dev), torch.from_numpy(fz).to(dev), cbp["color_front"],
        cbp["mask_front"], cbp["color_back"], res=2048).float().cpu().numpy())
    out = timed("shear", M.shear_transformation, out)
    torch.cuda.synchronize()
    print("rep", rep, "total %.3f s" % (time.time() - t_all))
    for k, t in T.items():
        print("  %-18s %7.1f ms" % (k, t * 1e3))
