"""What the occupancy-grid refresh of every 16th step costs the NSR stage: the stage on the bench's
data path with the reference's schedule (n = 16) and with the refreshes after the warm-up switched
off (n = 10**9 from step 256 on: the grid then stays what it was; NOT the reference's training —
a timing probe only).  usage: occ_update_cost.py [nsr_steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
from drawingspinup_amd.drawing import DrawingPipeline, synthetic_drawing  # noqa: E402
from drawingspinup_amd.nsr import render  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
orig = render.OccupancyGrid.every_n_step
count = {"n": 0}


def patched(self, step, occ_eval_fn, *a, **k):
    if step >= 256 and patched.off:
        return
    if step % 16 == 0:
        count["n"] += 1
    return orig(self, step, occ_eval_fn, *a, **k)


patched.off = False
render.OccupancyGrid.every_n_step = patched
out = {}
for off in (False, True, False, True):
    patched.off = off
    count["n"] = 0
    pipe = DrawingPipeline(dev, seed=0, mv_steps=3, nsr_steps=steps, n_frames=1, with_contour=False)
    pipe.time_substages = True
    drawing = synthetic_drawing(0, device=dev)
    normals, colors = pipe.multiview(drawing, 123456)
    torch.cuda.synchronize()
    t = time.time()
    pipe.reconstruct(normals, colors, drawing, 123456)
    torch.cuda.synchronize()
    out.setdefault("no_refresh_after_256" if off else "reference_schedule", []).append(
        {"ms_per_step": round(pipe.substage_seconds["nsr_fit"] / steps * 1e3, 4), "refreshes": count["n"]})
print(json.dumps(out))
