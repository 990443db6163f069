import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drawingspinup_amd.style.generators import build_model
from drawingspinup_amd.drawing import STYLE_ARGS
dev='cuda'
torch.manual_seed(0)
g1 = build_model("GeneratorJ_RIC", STYLE_ARGS, dev).eval()
g2 = build_model("GeneratorJ", STYLE_ARGS, dev).eval()
x = torch.rand(1,6,512,512,device=dev)*2-1
with torch.no_grad():
    for _ in range(2):
        y1 = g1(x); y2 = g2(x)
torch.cuda.synchronize()
import time
t=time.time()
with torch.no_grad():
    for _ in range(3):
        y1 = g1(x); y2 = g2(x)
torch.cuda.synchronize(); print("frame ms", (time.time()-t)/3*1e3)
