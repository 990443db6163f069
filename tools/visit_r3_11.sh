#!/bin/bash
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v11; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 200 python tools/nsr_stage_ab.py 3000 2>$O/err_$name.txt | tail -1 > $O/ab_$name.txt; echo "== $name $*"; cat $O/ab_$name.txt; }
run warm X=1
run all X=1
run m4 DSU_SC_MERGE_LEVELS=4
run m3 DSU_SC_MERGE_LEVELS=3
run m2 DSU_SC_MERGE_LEVELS=2
run m0 DSU_SC_MERGE_LEVELS=0
DSU_SC_MERGE_LEVELS=2 timeout 300 python -m pytest tests/test_gpu_hashgrid.py -q -m gpu 2>&1 | tail -2
