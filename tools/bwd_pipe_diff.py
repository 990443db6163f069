"""Where sdf_fd_bwd_pipe_kernel's MLP gradients differ from the general kernel's (debug probe)."""
import sys
import torch
sys.path.insert(0, "tests")
from drawingspinup_amd import ops
import test_gpu_hashgrid as t

dev = torch.device("cuda:0")
CFG = t.CFG
for active in (4, 5, 7):
    for n in (1, 2, 33, 64, 65, 200, 64 * 256 + 5, 256 * 96 + 40, 256 * 288 + 17, 70001):
        tab = t._table(61, 0.5).to(dev)
        mlp = [m.to(dev) for m in t._mlp(62)]
        pts = t._pts(n, 63 + n % 7, -1.0, 1.0).to(dev)
        eps, radius = 1.0 / 128, 1.0
        g = torch.Generator().manual_seed(64)
        d = [torch.randn(n, generator=g).to(dev), (torch.randn(n, 3, generator=g) * 0.1).to(dev),
             (torch.randn(n, 13, generator=g) * 0.1).to(dev), (torch.randn(n, generator=g) * 1e-4).to(dev)]
        fwd = ops.sdf_fd_fwd(CFG, tab, mlp, pts, radius, eps, active, enc_cache=True)
        gt0, gm0 = ops.sdf_fd_bwd(CFG, tab, mlp, pts, radius, eps, active, *d)
        gt1, gm1 = ops.sdf_fd_bwd(CFG, tab, mlp, pts, radius, eps, active, *d, enc_cache=fwd[4])
        line = f"active {active} n {n}:"
        for k, (a, b) in enumerate(zip(gm0, gm1)):
            df = (a - b).abs()
            nz = (a != b).nonzero()
            line += f" | g{k} {tuple(a.shape)} ndiff {len(nz)} max {float(df.max()):.3e} rel {float(df.max() / a.abs().max().clamp_min(1e-30)):.2e}"
            if 0 < len(nz) <= 6:
                line += " at " + str(nz.tolist())
            elif len(nz):
                cols = sorted(set(nz[:, -1].tolist()))
                line += f" cols {cols[:12]}{'...' if len(cols) > 12 else ''}"
        print(line, flush=True)
