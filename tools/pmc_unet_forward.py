import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drawingspinup_amd.mv.pipeline import build_random_pipeline
pipe = build_random_pipeline()
unet = pipe.unet
x = torch.randn(12,8,32,32,device='cuda').half(); ctx=torch.randn(12,1,768,device='cuda').half(); cl=torch.randn(12,10,device='cuda').half()
ts = torch.tensor([500],device='cuda')
for _ in range(3): unet(x, ts, ctx, cl)
torch.cuda.synchronize()
