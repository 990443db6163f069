#!/bin/bash
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v17; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_nsr_step.py tests/test_gpu_nsr_native.py tests/test_gpu_nsr_model.py -q -m gpu 2>&1 | grep -v Warning | tail -6
run() { name=$1; shift; env "$@" timeout 200 python tools/nsr_stage_ab.py 1500 2>$O/err_$name.txt | tail -1 > $O/ab_$name.txt; echo "== $name $*"; cat $O/ab_$name.txt; }
run warm X=1
run gate1 X=1
run gate2 DSU_NSR_PACK_GATE=2
run gate1b X=1
run gate2b DSU_NSR_PACK_GATE=2
cd /tmp && rm -rf /tmp/tr && DSU_NSR_PACK_GATE=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/tools/nsr_stage_ab.py 400 > /dev/null 2>&1; cd $R
f=$(find /tmp/tr -name '*kernel_trace.csv' | head -1)
python tools/trace_step_timeline.py "$f" > $O/timeline_gate2.txt 2>&1; head -24 $O/timeline_gate2.txt; grep -n "idle gaps > 4" -A8 $O/timeline_gate2.txt
