#!/bin/bash
# round-3 measurement visit: full GPU suite, the default bench line, the same under rocprofv3 --stats,
# step timeline with gap attribution
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v10; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v Warning | tail -15 > $O/pytest_gpu_tail.txt; tail -5 $O/pytest_gpu_tail.txt
timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-900 $O/bench.json
bash tools/gpu_rocprof_bench.sh r3v10_prof --no-cpu-baseline > $O/rocprof.log 2>&1; ls gpurun_out/r3v10_prof | head
cd /tmp && rm -rf /tmp/tr && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/tools/nsr_stage_ab.py 400 > /dev/null 2>&1; cd $R
f=$(find /tmp/tr -name '*kernel_trace.csv' | head -1)
python tools/trace_step_timeline.py "$f" > $O/timeline.txt 2>&1; head -8 $O/timeline.txt; grep -n "idle gaps > 4" -A14 $O/timeline.txt
