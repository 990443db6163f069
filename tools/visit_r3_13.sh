#!/bin/bash
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v13; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_nsr_native.py -q -m gpu 2>&1 | grep -v Warning | tail -12
run() { name=$1; shift; env "$@" timeout 200 python tools/nsr_stage_ab.py 3000 2>$O/err_$name.txt | tail -1 > $O/ab_$name.txt; echo "== $name $*"; cat $O/ab_$name.txt; }
run warm X=1
run coherent1 X=1
run coherent0 DSU_NSR_COHERENT=0
run coherent1b X=1
cd /tmp && rm -rf /tmp/tr && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/tools/nsr_stage_ab.py 400 > /dev/null 2>&1; cd $R
f=$(find /tmp/tr -name '*kernel_trace.csv' | head -1)
python tools/trace_step_timeline.py "$f" > $O/timeline.txt 2>&1; head -24 $O/timeline.txt
