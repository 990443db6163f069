"""The two modes of the NSR stage: N reconstructions back to back in one process (same drawing, same seed),
per drawing the HIP-event brackets of the geometry families and the stage's wall time.  Under
`rocprofv3 --kernel-trace` tools/nsr_modes_trace.py splits the trace into the drawings (3000 forward launches each).
usage: nsr_modes_probe.py [n_drawings] [nsr_steps] [side stream priority: 1 high (default) / 2 normal / 0 low] [side stream pool: 0 / 1]"""
import os, sys, time, json, torch
sys.path.insert(0, os.getcwd())
import bench
from drawingspinup_amd.drawing import DrawingPipeline, synthetic_drawing
from drawingspinup_amd.nsr import system as nsr_system
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
if len(sys.argv) > 3:
    from drawingspinup_amd import _lib
    _lib.check(_lib.lib().dsu_set_nsr_side_stream_priority(int(sys.argv[3])), "dsu_set_nsr_side_stream_priority")
    if len(sys.argv) > 4:
        _lib.check(_lib.lib().dsu_set_nsr_side_stream_pooling(int(sys.argv[4])), "dsu_set_nsr_side_stream_pooling")
timer = bench.KernelTimer(7); timer.install()
pipe = DrawingPipeline(dev, seed=0, mv_steps=3, nsr_steps=steps, n_frames=1, with_contour=False)
pipe.time_substages = True
drawing = synthetic_drawing(0, device=dev)
normals, colors = pipe.multiview(drawing, 123456)
torch.cuda.synchronize()
for k in range(n):
    timer.fam, timer._calls, nsr_system.native_timing["totals"] = {}, {}, {}
    timer.enabled = True
    if k == n - 1:
        torch.cuda.empty_cache()
    t = time.time()
    pipe.reconstruct(normals, colors, drawing, 123456)
    torch.cuda.synchronize()
    tot = time.time() - t
    timer.enabled = False
    rows = {r["kernel"]: round(r["avg_launch_ms"], 4) for r in timer.summary()}
    print(json.dumps({"drawing": k, "nsr_s": round(tot, 3), "fit_s": round(pipe.substage_seconds["nsr_fit"], 3),
                      "mem_reserved_MB": torch.cuda.memory_reserved() >> 20, **rows}), flush=True)
