#!/bin/bash
# full GPU test suite + the bench line the driver's command produces (more steps than the default)
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/${1:-full}; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -v Warning | tail -30 > $O/pytest_gpu_tail.txt; tail -8 $O/pytest_gpu_tail.txt
timeout 1500 python bench.py --steps ${STEPS:-6} --warmup ${WARMUP:-1} 2>$O/bench.err | tail -1 > $O/bench.json; cat $O/bench.json | cut -c1-400
