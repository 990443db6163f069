"""Timeline of the NSR optimisation step from a rocprofv3 kernel trace (csv): per queue the busy
time and the idle gaps between consecutive kernels, and one steady-state step printed kernel by
kernel (start offset, duration, queue).  usage: trace_step_timeline.py <kernel_trace.csv> [anchor]
`anchor` = substring of the kernel that starts a step (default sdf_fd_fwd: any form of the geometry forward)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "sdf_fd_fwd"
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"],
                 r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:]))
rows.sort()
starts = [i for i, r in enumerate(rows) if anchor in r[3]]
if len(starts) < 40:
    raise SystemExit("too few steps in the trace")
lo, hi = starts[len(starts) // 2], starts[len(starts) // 2 + 30]
win = rows[lo:hi]
t0, t1 = win[0][0], rows[hi][0]
steps = 30
print(f"window: {steps} steps, {(t1 - t0) / steps / 1e3:.1f} us per step")
byq = defaultdict(list)
for r in win:
    byq[r[2]].append(r)
for q, rs in byq.items():
    busy = sum(e - s for s, e, _, _ in rs)
    gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"queue {q}: {len(rs) / steps:.1f} kernels/step, busy {busy / steps / 1e3:.1f} us/step, "
          f"idle gaps {sum(pos) / steps / 1e3:.1f} us/step (max {max(pos) / 1e3 if pos else 0:.1f} us)")
agg = defaultdict(lambda: [0, 0])
for s, e, q, n in win:
    agg[(q, n)][0] += 1
    agg[(q, n)][1] += e - s
print("per kernel (queue, name): launches/step, us/step")
for (q, n), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  q{q} {n:42s} {c / steps:5.1f} {t / steps / 1e3:8.1f}")
one = rows[starts[len(starts) // 2 + 5]:starts[len(starts) // 2 + 6]]
b = one[0][0]
print("one step:")
for s, e, q, n in one:
    print(f"  +{(s - b) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  q{q}  {n}")

# the longest step of the window (the occupancy refresh of every 16th step: its samples cannot be
# prefetched), all queues
base = len(starts) // 2
longest = max(range(base, base + 30), key=lambda i: rows[starts[i + 1]][0] - rows[starts[i]][0])
# the refresh follows the step's optimizer launch: print from there to the next forward
seg = rows[starts[longest]:starts[longest + 1] + 1]
cut = max((i for i, r in enumerate(seg) if "small_update" in r[3]), default=0)
seg = seg[cut:]
b = seg[0][0]
print(f"longest step of the window: {(rows[starts[longest + 1]][0] - rows[starts[longest]][0]) / 1e3:.1f} us; "
      f"kernels between the previous optimizer launch and the next forward:")
for s_, e_, q, n in seg:
    print(f"  +{(s_ - b) / 1e3:8.1f} us  {(e_ - s_) / 1e3:7.1f} us  q{q}  {n}")

# where the main queue idles: gaps > 4 us between consecutive kernels of the busiest queue, by
# (kernel before -> kernel after), summed over the window
mainq = max(byq, key=lambda q: sum(e - s for s, e, _, _ in byq[q]))
rs = byq[mainq]
gap = defaultdict(lambda: [0, 0])
for i in range(len(rs) - 1):
    g = rs[i + 1][0] - rs[i][1]
    if g > 4000:
        k = (rs[i][3][-28:], rs[i + 1][3][-28:])
        gap[k][0] += 1
        gap[k][1] += g
print("idle gaps > 4 us on the main queue (before -> after): count/step, us/step")
for k, (c, t) in sorted(gap.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {k[0]:>28s} -> {k[1]:<28s} {c / steps:5.2f} {t / steps / 1e3:7.1f}")
