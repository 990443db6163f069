#!/bin/bash
# rocprofv3 --kernel-trace --stats of the NSR stage on the bench's data path (tools/nsr_stage_ab.py):
# top kernels by total time -> gpurun_out/<tag>/nsr_stage_kernel_stats.txt
tag=${1:-nsrtrace}; steps=${2:-3000}
export TMPDIR=/tmp PYTHONPATH=$(pwd)
out=gpurun_out/$tag; mkdir -p $out
w=/tmp/nsrtrace_$tag; rm -rf $w
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $w -o t -- python tools/nsr_stage_ab.py $steps > $out/trace.log 2>&1
f=$(find $w -name '*kernel_stats.csv' | head -1)
python - "$f" > $out/nsr_stage_kernel_stats.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:28]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-52s calls %6s avg %8.1f us  total %8.1f ms  %5.1f %%" % (n[:52], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
P
tail -2 $out/trace.log; cat $out/nsr_stage_kernel_stats.txt
rm -rf $w
