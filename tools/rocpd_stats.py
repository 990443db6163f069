"""Per-kernel statistics from a rocprofv3 (ROCm 7.2) sqlite result: python tools/rocpd_stats.py x_results.db [top]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), min(start), max(end) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
span = max(r[7] for r in rows) - min(r[6] for r in rows)
print(f"# total kernel time {tot/1e6:.2f} ms over a span of {span/1e6:.2f} ms ({len(rows)} kernels)")
print("calls,total_ms,avg_us,min_us,max_us,pct,name")
for r in rows[:top]:
    print(f"{r[1]},{r[2]/1e6:.3f},{r[3]/1e3:.2f},{r[4]/1e3:.2f},{r[5]/1e3:.2f},{100*r[2]/tot:.2f},{r[0][:110]}")
