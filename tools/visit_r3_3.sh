#!/bin/bash
# round-3 visit 3: failing-test details, scatter-kernel prefetch A/B, kernel-trace timeline of the native step
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_nsr_native.py tests/test_gpu_nsr_reference_step.py tests/test_gpu_hashgrid.py tests/test_gpu_nsr_step.py -q -m gpu -s 2>&1 | grep -v Warning | tail -120 > $O/tests.txt; tail -60 $O/tests.txt
timeout 200 python tools/nsr_stage_ab.py 1500 2>/dev/null | tail -1 > $O/ab_native.txt; cat $O/ab_native.txt
cd /tmp && rm -rf /tmp/tr && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/tools/nsr_stage_ab.py 400 > /dev/null 2>&1; cd $R
f=$(find /tmp/tr -name '*kernel_trace.csv' | head -1)
python tools/trace_step_timeline.py "$f" > $O/timeline_native.txt 2>&1; cat $O/timeline_native.txt
