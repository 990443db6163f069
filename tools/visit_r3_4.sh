#!/bin/bash
# round-3 visit 4: native-driver tests, A/B of the prefetch placement / priority / rays per marching wave
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_nsr_native.py tests/test_gpu_nsr_reference_step.py tests/test_gpu_render.py -q -m gpu -s 2>&1 | grep -v Warning | tail -60 > $O/tests.txt; tail -25 $O/tests.txt
run() { name=$1; shift; env "$@" timeout 200 python tools/nsr_stage_ab.py 1500 2>/dev/null | tail -1 > $O/ab_$name.txt; echo "== $name $*"; cat $O/ab_$name.txt; }
run default X=1
run packgate0 DSU_NSR_PACK_GATE=0
run prio0 DSU_NSR_SIDE_PRIO=0
run threads32 DSU_MARCH_THREADS=32
run threads16 DSU_MARCH_THREADS=16
run threads16_prio0 DSU_MARCH_THREADS=16 DSU_NSR_SIDE_PRIO=0
cd /tmp && rm -rf /tmp/tr && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/tools/nsr_stage_ab.py 400 > /dev/null 2>&1; cd $R
f=$(find /tmp/tr -name '*kernel_trace.csv' | head -1)
python tools/trace_step_timeline.py "$f" > $O/timeline_default.txt 2>&1; head -24 $O/timeline_default.txt; grep -n "one step" -A45 $O/timeline_default.txt
