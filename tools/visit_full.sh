#!/bin/bash
# full GPU test suite + the default bench line (what the driver runs at round end)
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/${1:-full}; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v Warning | tail -30 > $O/pytest_gpu_tail.txt; tail -8 $O/pytest_gpu_tail.txt
timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json; cat $O/bench.json | cut -c1-1500
