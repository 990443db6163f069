"""NSR stage A/B on the bench's own data path: 3 DDIM steps -> reconstruct (N steps), HIP-event
timing of the geometry forward / backward families.  usage: nsr_ab.py [nsr_steps]"""
import os, sys, time, json, torch
sys.path.insert(0, os.getcwd())
import bench
from drawingspinup_amd.drawing import DrawingPipeline, synthetic_drawing
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
timer = bench.KernelTimer(int(os.environ.get('NSR_AB_STRIDE', '7'))); timer.install()
pipe = DrawingPipeline(dev, seed=0, mv_steps=3, nsr_steps=steps, n_frames=1, with_contour=False)
pipe.time_substages = True
drawing = synthetic_drawing(0, device=dev)
normals, colors = pipe.multiview(drawing, 123456)
torch.cuda.synchronize()
timer.enabled = os.environ.get("NSR_AB_TIMING", "1") != "0"     # 0: no HIP-event marks around the geometry families
t = time.time()
pipe.reconstruct(normals, colors, drawing, 123456)
torch.cuda.synchronize()
tot = time.time() - t
rows = {r["kernel"]: round(r["avg_launch_ms"], 4) for r in timer.summary()}
print(json.dumps({"lib": os.path.basename(os.environ.get("DSU_HIP_LIB", "default")), "nsr_s": round(tot, 3),
                  "ms_per_step": round(pipe.substage_seconds["nsr_fit"] / steps * 1e3, 4),
                  "export_s": round(pipe.substage_seconds["nsr_export"], 3), **rows}))
