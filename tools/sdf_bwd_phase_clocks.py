"""Per-phase shader-clock totals of the geometry backward (needs the `prof` variant:
python -m drawingspinup_amd.build --variant prof -DDSU_BWD_PROF; run with
DSU_HIP_LIB=drawingspinup_amd/variants/libdsu_hip_prof.so [DSU_BWD_SPLIT=1])."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drawingspinup_amd import ops
from drawingspinup_amd._lib import lib
dev = 'cuda'
cfg = ops.HashGridConfig()
g = torch.Generator().manual_seed(0)
tab = ((torch.rand(cfg.n_entries, 2, generator=g) * 2 - 1) * 0.1).half().to(dev)
mlp = [(torch.randn(64, 23, generator=g) * 0.3).to(dev), (torch.randn(64, generator=g) * 0.05).to(dev),
       (torch.randn(13, 64, generator=g) * 0.2).to(dev), (torch.randn(13, generator=g) * 0.1).to(dev)]
N = 262144
r = torch.rand(2048, 2, generator=g) * 1.0 - 0.5
t = (torch.arange(128) - 64) * 3.383e-3
pts = torch.cat([r[:, None, :].expand(-1, 128, -1), t[None, :, None].expand(2048, -1, 1)], -1).reshape(-1, 3).contiguous().to(dev)
d = [torch.randn(N, device=dev), torch.randn(N, 3, device=dev), torch.randn(N, 13, device=dev), torch.randn(N, device=dev) * 1e-3]
gt = torch.zeros(cfg.n_params, device=dev)
names = ["pos+cache row", "upstream grads", "partner shuffles", "layer0+softplus", "dPre/sigmoid/dIn", "staging",
         "gW0 GEMM", "gW1", "dIn out / scatter", "flush", "tail"]
buf = (C.c_ulonglong * 16)()
for act in (4, 6):
    cache = ops.sdf_fd_fwd(cfg, tab, mlp, pts, 1.0, 0.02, act, True, True, True, enc_cache=True)[-1]
    for _ in range(3): ops.sdf_fd_bwd(cfg, tab, mlp, pts, 1.0, 0.02, act, *d, grad_table=gt, enc_cache=cache)
    torch.cuda.synchronize(); lib().dsu_debug_bwd_prof(buf, 1)
    reps = 10
    for _ in range(reps): ops.sdf_fd_bwd(cfg, tab, mlp, pts, 1.0, 0.02, act, *d, grad_table=gt, enc_cache=cache)
    torch.cuda.synchronize(); lib().dsu_debug_bwd_prof(buf, 1)
    tot = sum(buf[:11])
    waves = 512 * 4   # resident workgroups x waves
    print(f"active {act}: total {tot / reps / 1024 / 1e3:.1f} kclk per SIMD-slot per launch (1024 concurrent waves)")
    for k, nm in enumerate(names):
        print(f"   {nm:20s} {buf[k] / tot * 100:5.1f} %   {buf[k] / reps / 1024 / 1e3:8.1f} kclk")
