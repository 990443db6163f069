"""UNet evaluation inside the denoising loop: eager launches vs replay of the captured HIP graph
(MVDiffusionImagePipeline._unet_step, DSU_MV_GRAPH).  ms per evaluation, and output equality."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drawingspinup_amd.mv.pipeline import build_random_pipeline
pipe = build_random_pipeline()
dev = pipe.device
g = torch.Generator().manual_seed(1)
x = torch.randn(12, 8, 32, 32, generator=g).half().to(dev)
ctx = torch.randn(12, 1, 768, generator=g).half().to(dev)
cl = torch.randn(12, 10, generator=g).half().to(dev)
t = torch.tensor([500], device=dev)
outs = {}
with torch.no_grad():
    for mode in (False, True, False, True):
        pipe.use_graph = mode
        t0 = time.time()
        for _ in range(3): o = pipe._unet_step(x, t, ctx, cl)
        torch.cuda.synchronize(); first = time.time() - t0
        t0 = time.time()
        for _ in range(40): o = pipe._unet_step(x, t, ctx, cl)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / 40 * 1e3
        outs[mode] = o.clone()
        print(f"graph={mode}: {ms:.2f} ms per evaluation (first 3 calls {first:.2f} s)")
print("bit-identical:", bool(torch.equal(outs[False], outs[True])))
