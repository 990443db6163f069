#!/bin/bash
# which kernels differ between the slow and the fast NSR drawings of one process: event brackets + kernel trace per drawing
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_modes}; mkdir -p $O
w=/tmp/modes_trace; rm -rf $w
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $w -o t -- python tools/nsr_modes_probe.py 4 3000 2>/dev/null | grep '^{' | tee $O/brackets.txt
f=$(find $w -name '*kernel_trace.csv' | head -1); head -2 "$f" | cut -c1-400 > $O/trace_head.txt
python tools/nsr_modes_trace.py "$f" 3000 | tee $O/per_drawing_kernels.txt
rm -rf $w
