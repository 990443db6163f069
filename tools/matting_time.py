"""IS-Net forward at 1024^2 (mv/matting.py): total per image and the time by operation class
(HIP-event pairs around every module call), exact-f32 convolutions vs bf16 x 3.
    python tools/matting_time.py"""
import os, sys, time, collections, torch
sys.path.insert(0, os.getcwd())
from drawingspinup_amd.mv import matting
import torch.nn.functional as F
dev = torch.device("cuda:0")
net = matting.load_isnet(None, dev)
x = torch.rand(1, 3, 1024, 1024, device=dev) - 0.5
for mode in ("x3", "f32"):
    matting.EVAL_X3 = mode == "x3"
    with torch.no_grad():
        for _ in range(2):
            net(x)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(5):
            net(x)
        torch.cuda.synchronize()
    print(f"{mode}: {(time.time() - t) / 5 * 1e3:.1f} ms per 1024^2 image")
x4 = torch.rand(4, 3, 1024, 1024, device=dev) - 0.5
for mode in ("x3", "f32"):
    matting.EVAL_X3 = mode == "x3"
    with torch.no_grad():
        net(x4); torch.cuda.synchronize(); t = time.time()
        for _ in range(3):
            net(x4)
        torch.cuda.synchronize()
    print(f"{mode}, batch of 4: {(time.time() - t) / 3 * 1e3:.1f} ms per 4 images")
matting.EVAL_X3 = False
# by class
acc = collections.defaultdict(float)
def wrap(mod_name, obj, name):
    orig = getattr(obj, name)
    def f(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); out = orig(*a, **k); e.record(); ev.append((mod_name, s, e)); return out
    setattr(obj, name, f)
ev = []
wrap("conv_rebn", matting, "_rebnconv_hip"); wrap("conv_plain", matting, "_conv_hip")
wrap("upsample", matting, "_upsample_like"); wrap("max_pool", F, "max_pool2d"); wrap("cat", torch, "cat")
with torch.no_grad():
    torch.cuda.synchronize(); t = time.time(); net(x); torch.cuda.synchronize(); tot = time.time() - t
for n, s, e in ev:
    acc[n] += s.elapsed_time(e)
print("instrumented total %.1f ms;" % (tot * 1e3), {k: round(v, 2) for k, v in acc.items()}, "calls", collections.Counter(n for n, _, _ in ev))
