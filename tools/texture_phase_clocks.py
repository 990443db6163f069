"""Per-phase shader-clock totals of texture_bwd_kernel (needs the `texprof` variant:
python -m drawingspinup_amd.build --variant texprof -DDSU_TEX_PROF=1; run with
DSU_HIP_LIB=drawingspinup_amd/variants/libdsu_hip_texprof.so)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from drawingspinup_amd import ops
from drawingspinup_amd._lib import lib
dev = "cuda"
n = 262144
g = torch.Generator().manual_seed(0)
P = [torch.randn(64, 16, generator=g) * 0.4, torch.randn(64, generator=g) * 0.1, torch.randn(64, 64, generator=g) * 0.2,
     torch.randn(64, generator=g) * 0.1, torch.randn(3, 64, generator=g) * 0.3, torch.randn(3, generator=g) * 0.1]
Pd = [p.to(dev).contiguous() for p in P]
fd, gd = (torch.randn(n, 13, generator=g) * 0.5).to(dev), torch.randn(n, 3, generator=g).to(dev)
drd, dnd = (torch.randn(n, 3, generator=g) * 1e-3).to(dev), (torch.randn(n, 3, generator=g) * 1e-3).to(dev)
normal, rgb = ops.texture_fwd_shaded(Pd, fd, gd)
names = ["block head / rows", "forward recompute", "dPre1 (W2^T dz, relu')", "gW2 stage + GEMM", "gW1 stage + GEMM", "dH0 (bf16 x 3) + relu'",
         "gW0 stage + GEMM", "dIn (bf16 x 3)", "tail wait", "prologue / loop", "workgroup reduction", "write-out of a half"]
buf = (C.c_ulonglong * 16)()
f = lib().dsu_debug_tex_prof
f.argtypes = [C.c_void_p, C.c_int]
for _ in range(3): ops.texture_bwd_shaded_partials(Pd, fd, gd, rgb, drd, dnd, 4096)
torch.cuda.synchronize(); f(buf, 1)
reps = 10
for _ in range(reps): ops.texture_bwd_shaded_partials(Pd, fd, gd, rgb, drd, dnd, 4096)
torch.cuda.synchronize(); f(buf, 1)
tot = sum(buf[:12])
print(f"total {tot / reps / 1024 / 1e3:.1f} kclk per wave per launch (1024 waves, 8 halves each)")
for k, nm in enumerate(names):
    print(f"   {nm:26s} {buf[k] / tot * 100:5.1f} %   {buf[k] / reps / 1024 / 1e3:8.2f} kclk  ({buf[k] / reps / 1024 / 8:7.0f} clk per half)")
