"""Variant build only (-DDSU_AB_SWITCHES): how many workgroups of sdf_fd_scatter_kernel took the dense
tile / the cache per level over an NSR fit on the bench's data path, and the mean tile size.
    python -m drawingspinup_amd.build --variant ab -DDSU_AB_SWITCHES
    DSU_HIP_LIB=drawingspinup_amd/variants/libdsu_hip_ab.so python tools/scatter_stats.py [steps]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from drawingspinup_amd import _lib
from drawingspinup_amd.drawing import DrawingPipeline, synthetic_drawing
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
lib = _lib.lib()
assert lib.dsu_ab_switches(), "load the variant library through DSU_HIP_LIB"
raw = C.CDLL(_lib.LIB_PATH)
pipe = DrawingPipeline(dev, seed=0, mv_steps=3, nsr_steps=steps, n_frames=1, with_contour=False)
drawing = synthetic_drawing(0, device=dev)
normals, colors = pipe.multiview(drawing, 123456)
buf = (C.c_ulonglong * 48)()
raw.dsu_debug_sc_stats(buf, 1)
pipe.reconstruct(normals, colors, drawing, 123456)
torch.cuda.synchronize()
raw.dsu_debug_sc_stats(buf, 0)
for lev in range(10):
    d, c, v = buf[3 * lev], buf[3 * lev + 1], buf[3 * lev + 2]
    if d + c:
        print("level %d: dense %9d  cache %9d  (%.1f %% dense)  mean tile cells %.0f" % (lev, d, c, 100 * d / (d + c), v / max(d, 1)))

# clocks of the LAST launch per workgroup (wall_clock64: 100 MHz constant clock -> 10 ns units)
import numpy as np
clk = (C.c_ulonglong * 4096)()
raw.dsu_debug_sc_clocks(clk)
a = np.array(list(clk), dtype=np.float64).reshape(256, 16) * 0.01      # us
tot = a[:, 15]
print("last launch: workgroup total us: mean %.1f  median %.1f  p90 %.1f  max %.1f  min %.1f" %
      (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max(), tot.min()))
# per level: time the workgroups spent in items of that level during the last launch (mean over workgroups)
for lev in range(10):
    if a[:, lev].sum() > 0:
        print("   level %d: mean %.1f us per workgroup (%.1f %% of its total)" % (lev, a[:, lev].mean(), 100 * a[:, lev].sum() / tot.sum()))

