"""Texture MLP kernels on one step's sample count (n = 262 144): ms per launch of the forward and the
backward (shaded forms, as the NSR step calls them), and the backward's gradients against torch
autograd in float64 (relative L2 per tensor).   python tools/texture_time.py [n]"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from drawingspinup_amd import ops
dev = "cuda"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
g = torch.Generator().manual_seed(0)
P = [torch.randn(64, 16, generator=g) * 0.4, torch.randn(64, generator=g) * 0.1,
     torch.randn(64, 64, generator=g) * 0.2, torch.randn(64, generator=g) * 0.1,
     torch.randn(3, 64, generator=g) * 0.3, torch.randn(3, generator=g) * 0.1]
feat = torch.randn(n, 13, generator=g) * 0.5
grad = torch.randn(n, 3, generator=g)
d_rgb = torch.randn(n, 3, generator=g) * 1e-3
d_nrm = torch.randn(n, 3, generator=g) * 1e-3
Pd = [p.to(dev).contiguous() for p in P]
fd, gd, drd, dnd = feat.to(dev), grad.to(dev), d_rgb.to(dev), d_nrm.to(dev)


def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / reps


normal, rgb = ops.texture_fwd_shaded(Pd, fd, gd)
tf = timeit(lambda: ops.texture_fwd_shaded(Pd, fd, gd))
tb = timeit(lambda: ops.texture_bwd_shaded_partials(Pd, fd, gd, rgb, drd, dnd, 4096))
print(f"lib {os.path.basename(os.environ.get('DSU_HIP_LIB', 'default'))}: n {n}: texture fwd {tf * 1e3:.1f} us, bwd {tb * 1e3:.1f} us")
if os.environ.get("TEX_CHECK", "1") != "0":
    m = min(n, 20000)
    P64 = [p.double().requires_grad_(True) for p in P]
    f64, g64 = feat[:m].double().requires_grad_(True), grad[:m].double().requires_grad_(True)
    nrm = torch.nn.functional.normalize(g64, dim=-1)
    x = torch.cat([f64, nrm], -1)
    h = torch.relu(x @ P64[0].T + P64[1]); h = torch.relu(h @ P64[2].T + P64[3])
    out = torch.sigmoid(h @ P64[4].T + P64[5])
    (out * d_rgb[:m].double()).sum().backward(retain_graph=True)
    (nrm * d_nrm[:m].double()).sum().backward()
    dg, dfeat, gp = ops.texture_bwd_shaded(Pd, fd[:m].contiguous(), gd[:m].contiguous(), rgb[:m].contiguous(),
                                           drd[:m].contiguous(), dnd[:m].contiguous(), 0)
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
    print("  rel-L2 vs float64 autograd: d_feature %.2e d_grad %.2e" % (rel(dfeat, f64.grad), rel(dg, g64.grad)),
          " ".join("%s %.2e" % (k, rel(a, b.grad)) for k, a, b in zip(["w0", "b0", "w1", "b1", "w2", "b2"], gp, P64)))
