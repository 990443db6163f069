"""Geometry backward from the feature cache (the kernel the optimisation runs) against float64 autograd:
relative L2 and largest element error (in units of the tensor's largest entry) per gradient tensor."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from drawingspinup_amd import ops
import test_gpu_hashgrid as t
dev = torch.device("cuda:0")
for n, active in ((1037, 5), (20000, 4), (20000, 6)):
    tab = t._table(15, 0.5); mlp = t._mlp(16); pts = t._pts(n, 17, -1.0, 1.0)
    eps, radius = 0.031, 1.0
    g = torch.Generator().manual_seed(18)
    d = [torch.randn(n, generator=g), torch.randn(n, 3, generator=g) * 0.1, torch.randn(n, 13, generator=g), torch.randn(n, generator=g) * 1e-3]
    tab64 = tab.double().requires_grad_(True); mlp64 = [m.double().requires_grad_(True) for m in mlp]
    t._torch_fd_loss(tab64, mlp64, pts.numpy(), eps, active, radius, [x.double() for x in d]).backward()
    tabd, mlpd, ptsd = tab.to(dev), [m.to(dev) for m in mlp], pts.to(dev)
    fwd = ops.sdf_fd_fwd(t.CFG, tabd, mlpd, ptsd, radius, eps, active, enc_cache=True)
    gt, gm = ops.sdf_fd_bwd(t.CFG, tabd, mlpd, ptsd, radius, eps, active, *[x.to(dev) for x in d], enc_cache=fwd[4])
    out = []
    for name, got, ref in [("table", gt.cpu().double().reshape(-1, 2), tab64.grad)] + [(k, a.cpu().double(), b.grad) for k, a, b in zip(("w0", "b0", "w1", "b1"), gm, mlp64)]:
        out.append("%s relL2 %.2e maxerr/max %.2e" % (name, float((got - ref).norm() / ref.norm()), float((got - ref).abs().max() / ref.abs().max())))
    print(os.path.basename(os.environ.get("DSU_HIP_LIB", "default")), "n", n, "L", active, "|", " | ".join(out))
