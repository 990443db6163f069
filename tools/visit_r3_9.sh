#!/bin/bash
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v9; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 200 python tools/nsr_stage_ab.py 1500 2>$O/err_$name.txt | tail -1 > $O/ab_$name.txt; echo "== $name $*"; cat $O/ab_$name.txt; tail -2 $O/err_$name.txt | grep -i error; }
run warm X=1
run default X=1
run sc1024 DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_sc1024.so
run sc2wg DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_sc2wg.so
run sc256x4 DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_sc256x4.so
bash tools/pmc_sq_nsr.sh r3v9 300 2>&1 | grep -E "sdf_fd|texture_|ray_march" 
