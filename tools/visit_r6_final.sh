#!/bin/bash
# round-6 closing visit: full GPU suite, the bench line (default flags + more steps), rocprofv3 kernel
# statistics of the bench command, the driver's command (--steps 20 --warmup 5)
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/${1:-r6_final}; mkdir -p $O
V=drawingspinup_amd/variants
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v Warning | tail -30 > $O/pytest_gpu_tail.txt; tail -4 $O/pytest_gpu_tail.txt
timeout 900 python bench.py --steps 6 --warmup 1 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json
bash tools/gpu_rocprof_bench.sh $(basename $O)/rocprof --steps 2 --warmup 1
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_command.json; cut -c1-200 $O/bench_driver_command.json
