#!/bin/bash
# k1 load-wait restructuring: hash-grid / NSR tests, then the stage A/B (2 runs)
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3_25; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_nsr_native.py tests/test_gpu_nsr_reference_step.py tests/test_gpu_nsr_step.py -q -m gpu -x 2>&1 | grep -v Warn | tail -6 > $O/tests.txt; tail -3 $O/tests.txt
for i in 1 2; do timeout 200 python tools/nsr_stage_ab.py 1500 2>$O/err_$i.txt | tail -1 > $O/ab_$i.txt; cat $O/ab_$i.txt; done
