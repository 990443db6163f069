#!/bin/bash
# kernel-trace timeline of the NSR step (tools/trace_step_timeline.py) on the bench's data path:
# gpurun_out/<tag>/nsr_step_timeline.txt
tag=${1:-timeline}; steps=${2:-1500}
export TMPDIR=/tmp PYTHONPATH=$(pwd)
out=gpurun_out/$tag; mkdir -p $out
w=/tmp/tl_$tag; rm -rf $w
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $w -o t -- python tools/nsr_stage_ab.py $steps > $out/trace.log 2>&1
f=$(find $w -name '*kernel_trace.csv' | head -1)
python tools/trace_step_timeline.py "$f" > $out/nsr_step_timeline.txt 2>&1
# (the middle of the trace, for re-parsing off the box)
python - "$f" $out/kernel_trace_mid.csv <<'P'
import sys
l = open(sys.argv[1]).read().splitlines()
m = len(l) // 2
open(sys.argv[2], "w").write("\n".join([l[0]] + l[m:m + 4000]) + "\n")
P
head -45 $out/nsr_step_timeline.txt
rm -rf $w
