"""Geometry forward (dsu_sdf_fd_fwd_sorted) timed alone on fixed inputs, for A/B of variant
libraries (DSU_HIP_LIB=...), including the ablated ones of tools/ab_fwd_variants.sh.

    fwd_phase_probe.py capture FILE   run the NSR optimisation on the synthetic sphere with the
                                      Python-sequenced step and save the forward's real inputs
                                      (sorted samples + regulariser points, permutation, eps,
                                      table, effective weights) at steps 600 / 1600 / 2600
                                      (4 / 5 / 6 active levels)
    fwd_phase_probe.py time FILE [reps]   time the forward on those inputs
    fwd_phase_probe.py synthetic [reps]   2048 parallel rays x 130 samples (a thin slab: every
                                      gather hits the cache; isolates the arithmetic)
"""
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from drawingspinup_amd import ops
dev = torch.device("cuda:0")
cfg = ops.HashGridConfig()


def timeit(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return round(s.elapsed_time(e) / reps * 1e3, 1)


mode = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
out = {"lib": os.path.basename(os.environ.get("DSU_HIP_LIB", "default")), "mode": mode}
if mode == "capture":
    from drawingspinup_amd.nsr.system import OrthoData, OrthoNeuSSystem
    ds = OrthoData.synthetic_sphere(1024, device=dev)
    sysm = OrthoNeuSSystem(device=dev, seed=0)
    sysm.dataset = ds
    sysm.step_mode = "fused"
    saved, orig = {}, ops.sdf_fd_fwd
    want = {600: None, 1600: None, 2600: None}

    def hook(cfg_, tab, mlp, pts, radius, eps, active, *a, **k):
        step = int(sysm.global_step)
        if step in want and step not in saved and pts.shape[0] > 100000:
            saved[step] = {"tab": tab.clone(), "mlp": [m.clone() for m in mlp], "pts": pts.clone(),
                           "perm": None if k.get("perm") is None else k["perm"].clone(),
                           "radius": float(radius), "eps": float(eps), "active": int(active)}
        return orig(cfg_, tab, mlp, pts, radius, eps, active, *a, **k)
    ops.sdf_fd_fwd = hook
    for _ in range(2602):
        sysm.training_step()
    torch.save(saved, sys.argv[2])
    out["captured"] = {k: (int(v["pts"].shape[0]), v["active"], v["eps"]) for k, v in saved.items()}
elif mode == "time":
    saved = torch.load(sys.argv[2])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    g = torch.Generator().manual_seed(0)
    tab_s = ((torch.rand(cfg.n_entries, 2, generator=g) * 2 - 1) * 0.1).half().to(dev)
    mlp_s = [(torch.randn(64, 23, generator=g) * 0.3).to(dev), (torch.randn(64, generator=g) * 0.05).to(dev),
             (torch.randn(13, 64, generator=g) * 0.2).to(dev), (torch.randn(13, generator=g) * 0.1).to(dev)]
    for step, d in sorted(saved.items()):
        fn = lambda: ops.sdf_fd_fwd(cfg, d["tab"], d["mlp"], d["pts"], d["radius"], d["eps"], d["active"],
                                    True, True, False, enc_cache=True, perm=d["perm"])
        key = f"act{d['active']}_n{d['pts'].shape[0]}"
        out[key + "_us"] = timeit(fn, reps)
        r = fn()
        out[key + "_nan_sdf"] = int(torch.isnan(r[0]).sum())
        # which input makes the difference: real samples with random table / weights, and back
        out[key + "_us_random_weights"] = timeit(lambda: ops.sdf_fd_fwd(
            cfg, tab_s, mlp_s, d["pts"], d["radius"], d["eps"], d["active"], True, True, False,
            enc_cache=True, perm=d["perm"]), reps)
        out[key + "_us_random_table_only"] = timeit(lambda: ops.sdf_fd_fwd(
            cfg, tab_s, d["mlp"], d["pts"], d["radius"], d["eps"], d["active"], True, True, False,
            enc_cache=True, perm=d["perm"]), reps)
        out[key + "_us_unsorted_identity"] = timeit(lambda: ops.sdf_fd_fwd(
            cfg, d["tab"], d["mlp"], d["pts"], d["radius"], d["eps"], d["active"], True, True, False,
            enc_cache=True, perm=None), reps)
else:
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    g = torch.Generator().manual_seed(0)
    tab = ((torch.rand(cfg.n_entries, 2, generator=g) * 2 - 1) * 0.1).half().to(dev)
    mlp = [(torch.randn(64, 23, generator=g) * 0.3).to(dev), (torch.randn(64, generator=g) * 0.05).to(dev),
           (torch.randn(13, 64, generator=g) * 0.2).to(dev), (torch.randn(13, generator=g) * 0.1).to(dev)]
    R, S = 2048, 130
    xy = torch.rand(R, 2, generator=g) * 1.2 - 0.6
    t = torch.linspace(-0.22, 0.22, S)
    pts = torch.cat([xy[:, None, :].expand(-1, S, -1), t[None, :, None].expand(R, -1, 1)], -1).reshape(-1, 3)
    ps, perm = ops.spatial_sort(pts.contiguous().to(dev), 1.0)
    for act in (4, 5, 6):
        eps = 2.0 / (cfg.base_resolution * cfg.per_level_scale ** (act - 1))
        out[f"act{act}_us"] = timeit(lambda: ops.sdf_fd_fwd(cfg, tab, mlp, ps, 1.0, eps, act,
                                                            enc_cache=True, perm=perm), reps)
print(json.dumps(out))
