"""Geometry backward: time per active-level count with phases switched off (DSU_BWD_ABLATE bits:
1 scatter, 2 parameter GEMMs, 4 queue/cache, 8 flush atomics, 16 feature load, 32 softplus,
64 dIn MFMA).  Needs a library built WITHOUT -DDSU_NO_ABLATE."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drawingspinup_amd import ops
dev = 'cuda'
cfg = ops.HashGridConfig()
g = torch.Generator().manual_seed(0)
tab = ((torch.rand(cfg.n_entries, 2, generator=g) * 2 - 1) * 0.1).half().to(dev)
mlp = [(torch.randn(64, 23, generator=g) * 0.3).to(dev), (torch.randn(64, generator=g) * 0.05).to(dev),
       (torch.randn(13, 64, generator=g) * 0.2).to(dev), (torch.randn(13, generator=g) * 0.1).to(dev)]
N = 262144
r = torch.rand(2048, 2, generator=g) * 1.0 - 0.5
# 128 samples per ray at the training step size (render_step_size 3.383e-3 of the radius-1 box)
STEP = float(os.environ.get('PROBE_STEP', '3.383e-3'))
t = (torch.arange(128) - 64) * STEP
pts = torch.cat([r[:, None, :].expand(-1, 128, -1), t[None, :, None].expand(2048, -1, 1)], -1).reshape(-1, 3).contiguous().to(dev)
d = [torch.randn(N, device=dev), torch.randn(N, 3, device=dev), torch.randn(N, 13, device=dev), torch.randn(N, device=dev) * 1e-3]
gt = torch.zeros(cfg.n_params, device=dev)


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n


bits = [int(b) for b in sys.argv[1:]] or [0, 1, 2, 4, 8, 5, 13, 15, 32, 64, 96, 111, 127]
for act in (4, 5, 6):
    # feature cache as in training (the forward pass writes it)
    cache = ops.sdf_fd_fwd(cfg, tab, mlp, pts, 1.0, 0.02, act, True, True, True, enc_cache=True)[-1]
    row = []
    for b in bits:
        os.environ["DSU_BWD_ABLATE"] = str(b)
        row.append((b, timeit(lambda: ops.sdf_fd_bwd(cfg, tab, mlp, pts, 1.0, 0.02, act, *d, grad_table=gt,
                                                     enc_cache=cache))))
    print("active", act, " ".join(f"[{b}] {ms:.3f}" for b, ms in row), flush=True)
os.environ["DSU_BWD_ABLATE"] = "0"
