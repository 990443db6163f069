"""Where the part of DrawingPipeline.reconstruct BEFORE the optimisation goes (bench.py's nsr stage
minus matting / fit / export / post): upsampling, side masks, dataset, system construction, first step.
    python tools/recon_prefit_profile.py"""
import os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from drawingspinup_amd.drawing import DrawingPipeline, synthetic_drawing, fill_holes
from drawingspinup_amd.nsr.system import OrthoData, OrthoNeuSSystem
dev = torch.device("cuda:0")
pipe = DrawingPipeline(dev, seed=0, mv_steps=2, nsr_steps=10, n_frames=1, with_contour=False)
drawing = synthetic_drawing(0, device=dev)
normals, colors = pipe.multiview(drawing, 1)
T = {}
def timed(name, fn):
    torch.cuda.synchronize(); t = time.time(); out = fn(); torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + time.time() - t
    return out
for rep in range(3):
    T.clear()
    up = lambda t: F.interpolate(t.float(), size=(1024, 1024), mode="bicubic", align_corners=False).clamp(0, 1)
    col = timed("upsample", lambda: (up(colors).permute(0, 2, 3, 1), up(normals).permute(0, 2, 3, 1) * 2 - 1))[0]
    side = timed("threshold", lambda: (1.0 - col).amax(-1) > 12.0 / 255.0)
    filled = timed("fill_holes_x6", lambda: torch.stack([fill_holes(m) for m in side]))
    sysm = timed("system_ctor", lambda: OrthoNeuSSystem(device=dev, seed=rep))
    ds = timed("synthetic_dataset", lambda: OrthoData.synthetic_sphere(1024, device=dev))
    sysm.dataset = ds
    timed("first_step", lambda: sysm.training_step())
    timed("steps_2_to_10", lambda: [sysm.training_step() for _ in range(9)])
    print("rep", rep, {k: round(v * 1e3, 1) for k, v in T.items()})
t = time.time(); pipe.reconstruct(normals, colors, drawing, 5); torch.cuda.synchronize()
print("reconstruct(10 steps) total %.3f s; substages %s" % (time.time() - t, {k: round(v, 3) for k, v in pipe.substage_seconds.items()}))
