#!/bin/bash
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v12; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_nsr_step.py tests/test_gpu_nsr_reference_step.py -q -m gpu 2>&1 | grep -v Warning | tail -12
run() { name=$1; shift; env "$@" timeout 200 python tools/nsr_stage_ab.py 3000 2>$O/err_$name.txt | tail -1 > $O/ab_$name.txt; echo "== $name $*"; cat $O/ab_$name.txt; }
run warm X=1
run centre1 X=1
run centre0 DSU_SC_CENTRE=0
run centre1b X=1
