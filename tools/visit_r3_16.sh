#!/bin/bash
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v16; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_nsr_step.py tests/test_gpu_nsr_native.py -q -m gpu 2>&1 | grep -v Warning | tail -8
run() { name=$1; shift; env "$@" timeout 200 python tools/nsr_stage_ab.py "${STEPS:-1500}" 2>$O/err_$name.txt | tail -1 > $O/ab_$name.txt; echo "== $name $*"; cat $O/ab_$name.txt; }
run warm X=1
run default1500 X=1
STEPS=3000 run default3000 X=1
