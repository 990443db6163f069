#!/bin/bash
# remesh() of the export part by part + the kernel table of the same run
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_remesh}; mkdir -p $O
w=/tmp/remesh_prof; rm -rf $w
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $w -o t -- python tools/remesh_profile.py 512 3 2>&1 | grep -v "Warn\|rocprof\|^[EW]2026" > $O/remesh_profile.txt
f=$(find $w -name '*kernel_stats.csv' | head -1)
python - "$f" >> $O/remesh_profile.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# kernels of the whole run (mesh construction + %d x 2 remeshes): total %.1f ms" % (3, tot / 1e6))
for r in rows[:30]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-60s calls %6s avg %8.1f us  total %8.1f ms" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
P
cat $O/remesh_profile.txt
