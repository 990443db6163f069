#!/bin/bash
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v8; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_nsr_native.py tests/test_gpu_nsr_step.py tests/test_gpu_nsr_model.py -q -m gpu 2>&1 | grep -v Warning | tail -30 > $O/tests.txt; tail -6 $O/tests.txt
run() { name=$1; shift; env "$@" timeout 200 python tools/nsr_stage_ab.py 1500 2>/dev/null | tail -1 > $O/ab_$name.txt; echo "== $name $*"; cat $O/ab_$name.txt; }
run warm X=1
run default X=1
run sc1024 DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_sc1024.so
run sc2wg DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_sc2wg.so
run sc256x4 DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_sc256x4.so
run default2 X=1
