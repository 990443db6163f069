"""Where the export of one drawing goes ON THE BENCH'S DATA PATH (DrawingPipeline.reconstruct after a full
NSR fit of the synthetic drawing): every part of nsr/mesh.py's isosurface() wrapped with a
synchronise + wall clock (so the parts add up to more than the unsynchronised export of the bench),
plus the export as the bench times it.
    python tools/export_profile_bench.py [nsr_steps] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
from drawingspinup_amd.drawing import DrawingPipeline, synthetic_drawing  # noqa: E402
from drawingspinup_amd.nsr import mesh as M  # noqa: E402
from drawingspinup_amd.nsr import model as MODEL  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
T, CALLS = {}, {}
ON = [False]


def wrap(owner, name, label=None):
    fn = getattr(owner, name)
    label = label or name

    def timed(*a, **k):
        if not ON[0]:
            return fn(*a, **k)
        torch.cuda.synchronize()
        t0 = time.time()
        out = fn(*a, **k)
        torch.cuda.synchronize()
        T[label] = T.get(label, 0.0) + time.time() - t0
        CALLS[label] = CALLS.get(label, 0) + 1
        return out
    setattr(owner, name, timed)


for n in ("signed_distance_band", "smooth_constrained", "marching_cubes", "resize_cubic_u8", "remesh",
          "_remesh_host", "crop_front_mask", "vertex_colors"):
    wrap(M, n)
wrap(MODEL.NeuSModel, "isosurface_levels")
pipe = DrawingPipeline(dev, seed=0, mv_steps=3, nsr_steps=steps, n_frames=1, with_contour=False)
pipe.time_substages = True
drawing = synthetic_drawing(0, device=dev)
normals, colors = pipe.multiview(drawing, 123456)
for rep in range(reps):
    for on in (False, True):
        ON[0] = on
        T.clear(); CALLS.clear()
        pipe.reconstruct(normals, colors, drawing, 123456)
        ss = pipe.substage_seconds
        print(f"rep {rep} wrapped={on}: nsr_fit {ss['nsr_fit']:.3f} s, nsr_export {ss['nsr_export'] * 1e3:.1f} ms, "
              f"nsr_post {ss['nsr_post'] * 1e3:.1f} ms")
        if on:
            inner = T.get("signed_distance_band", 0.0)
            for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
                note = "  (inside smooth_constrained)" if k == "signed_distance_band" else (
                    "  (inside remesh)" if k == "_remesh_host" else "")
                print(f"  {k:24s} x{CALLS[k]}  {v * 1e3:8.1f} ms{note}")
            top = sum(v for k, v in T.items() if k not in ("signed_distance_band", "_remesh_host"))
            print(f"  sum of the outer parts     {top * 1e3:8.1f} ms; unaccounted {(ss['nsr_export'] - top) * 1e3:.1f} ms")
            print("  remesh stats", M.last_remesh_stats, "fine faces before remesh: see input_faces")
