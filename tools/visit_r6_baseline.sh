#!/bin/bash
# round-6 opening visit: suite, bench line, NSR stage kernel statistics, UNet per-op table, export timeline
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/${1:-r6_base}; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | grep -v Warning | tail -15 > $O/pytest_gpu_tail.txt; tail -4 $O/pytest_gpu_tail.txt
timeout 600 python bench.py --steps 3 --warmup 1 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-600 $O/bench.json
timeout 300 python tools/unet_op_times.py > $O/unet_op_times.txt 2>&1; tail -5 $O/unet_op_times.txt
timeout 300 python tools/export_profile.py 300 > $O/export_profile.txt 2>&1; tail -30 $O/export_profile.txt
bash tools/nsr_stage_trace.sh ${1:-r6_base} 3000 | tail -32
