"""GroupNorm(+SiLU) launch time per UNet shape (HIP events over 200 back-to-back calls)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drawingspinup_amd import ops
dev = "cuda"
for shp in [(12, 32, 32, 320), (12, 32, 32, 640), (12, 32, 32, 960), (12, 16, 16, 640), (12, 16, 16, 1280),
            (12, 8, 8, 1280), (12, 8, 8, 2560), (12, 4, 4, 1280)]:
    C = shp[-1]
    x = torch.randn(*shp, device=dev).half()
    w, b = torch.ones(C, device=dev).half(), torch.zeros(C, device=dev).half()
    for _ in range(10): ops.groupnorm_nhwc_f16(x, w, b, 32, 1e-5, True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(200): ops.groupnorm_nhwc_f16(x, w, b, 32, 1e-5, True)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 200 * 1e3
    print(f"{shp}: {us:7.1f} us  {2 * x.numel() * 2 / us / 1e6:6.2f} TB/s (read + write)")
