#!/bin/bash
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v18; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_nsr_step.py tests/test_gpu_nsr_native.py tests/test_gpu_nsr_reference_step.py tests/test_gpu_nsr_model.py -q -m gpu 2>&1 | grep -v Warning | tail -6
run() { name=$1; shift; env "$@" timeout 200 python tools/nsr_stage_ab.py 1500 2>$O/err_$name.txt | tail -1 > $O/ab_$name.txt; echo "== $name $*"; cat $O/ab_$name.txt; }
run warm X=1
run pair1 X=1
run pair0 DSU_FWD_PAIR=0
run pair1b X=1
run pair0b DSU_FWD_PAIR=0
