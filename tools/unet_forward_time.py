"""ms per UNet forward (B=12, 32x32 latents, full width) and the per-family HIP-event split."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drawingspinup_amd.mv.unet import UNetMV2DConditionModel
dev = 'cuda'
torch.manual_seed(0)
m = UNetMV2DConditionModel().half().to(dev).eval()
g = torch.Generator().manual_seed(1)
x = torch.randn(12, 8, 32, 32, generator=g).half().to(dev)
ctx = torch.randn(12, 1, 768, generator=g).half().to(dev)
cl = torch.randn(12, 10, generator=g).half().to(dev)
t = torch.tensor([500], device=dev)
with torch.no_grad():
    for _ in range(5): m(x, t, ctx, cl)
    torch.cuda.synchronize(); t0 = time.time()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    for _ in range(n): m(x, t, ctx, cl)
    torch.cuda.synchronize()
ms = (time.time() - t0) / n * 1e3
print(f"unet forward {ms:.2f} ms = {2.913 / ms * 1e3:.0f} TFLOP/s")
