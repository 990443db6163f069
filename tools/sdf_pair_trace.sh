#!/bin/bash
# per-kernel times of the geometry network's kernels on the PMC workload (N = 262144, 5 levels):
# rocprofv3 --kernel-trace --stats over tools/pmc_sdf_kernels.py -> gpurun_out/<tag>/sdf_pair_stats.csv
tag=${1:-sdfpair}
export TMPDIR=/tmp PYTHONPATH=$(pwd)
out=gpurun_out/$tag; mkdir -p $out
w=/tmp/sdfpair_$tag; rm -rf $w
PMC_ACTIVE=${2:-5} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $w -o t -- python tools/pmc_sdf_kernels.py > $out/trace.log 2>&1
f=$(find $w -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && grep -E "Name|sdf_fd|reduce_partials" "$f" | cut -c1-60,200-400 | sed 's/,/ /g' > $out/sdf_pair_stats.csv
python - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "sdf_fd" in n or "reduce_partials" in n:
        print("%-44s calls %4s avg %9.1f us  min %9.1f  max %9.1f" % (n.split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
P
rm -rf $w
