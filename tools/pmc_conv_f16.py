import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drawingspinup_amd import ops
dev='cuda'
g=torch.Generator().manual_seed(0)
def run(B,H,W,C,O,k):
    x=(torch.randn(B,H,W,C,generator=g)).half().to(dev)
    w=(torch.randn(O,k*k,C,generator=g)*0.02).half().to(dev)
    b=torch.zeros(O).half().to(dev)
    for _ in range(6): ops.conv2d_nhwc_f16(x,w,b,k,1,k//2)
    torch.cuda.synchronize()
run(12,32,32,640,320,3)
run(12,16,16,1280,640,3)
