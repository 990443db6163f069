"""Split a rocprofv3 kernel trace of tools/nsr_modes_probe.py into its drawings (3000 geometry-forward launches each) and
print the average duration of the step's kernels per drawing.  usage: nsr_modes_trace.py <kernel_trace.csv> [steps]"""
import csv, sys
from collections import defaultdict
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rows = list(csv.DictReader(open(sys.argv[1])))
key = lambda r: r["Kernel_Name"] if "Kernel_Name" in r else r["Name"]
st = lambda r: int(r.get("Start_Timestamp") or r.get("BeginNs"))
en = lambda r: int(r.get("End_Timestamp") or r.get("EndNs"))
fw = sorted((r for r in rows if "sdf_fd_fwd_shared" in key(r)), key=st)
bounds = [st(fw[i]) for i in range(0, len(fw), steps)] + [1 << 62]
names = ["sdf_fd_fwd_shared", "sdf_fd_bwd_pipe", "sdf_fd_scatter", "texture_fwd", "texture_bwd", "ray_march", "small_update"]
for d in range(len(bounds) - 1):
    acc = defaultdict(lambda: [0, 0])
    lo, hi = bounds[d], bounds[d + 1]
    for r in rows:
        s = st(r)
        if lo <= s < hi:
            for nme in names:
                if nme in key(r):
                    acc[nme][0] += en(r) - s; acc[nme][1] += 1
    print("drawing", d, {k: (round(v[0] / max(v[1], 1) / 1e3, 1), v[1]) for k, v in acc.items()})
