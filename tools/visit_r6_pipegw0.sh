#!/bin/bash
# geometry backward, pipelined MLP part: layer 0 of the recompute as bf16 x 3 (default) vs f32 (pf32: -DDSU_PIPE_L0_F32)
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_pipegw0}; mkdir -p $O
V=drawingspinup_amd/variants
timeout 900 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_nsr_reference_step.py tests/test_gpu_nsr_native.py tests/test_gpu_nsr_step.py tests/test_gpu_nsr_model.py -q -x 2>&1 | grep -v Warn | tail -8 | tee -a $O/ab.txt
for rep in 1 2; do
  DSU_HIP_LIB=$V/libdsu_hip_pf32.so python tools/pipe_fixed_cost.py 2>/dev/null | tail -1 | tee -a $O/ab.txt
  python tools/pipe_fixed_cost.py 2>/dev/null | tail -1 | tee -a $O/ab.txt
done
for rep in 1 2; do
  DSU_HIP_LIB=$V/libdsu_hip_pf32.so timeout 300 python tools/nsr_stage_ab.py 3000 2>/dev/null | tail -1 | tee -a $O/ab.txt
  timeout 300 python tools/nsr_stage_ab.py 3000 2>/dev/null | tail -1 | tee -a $O/ab.txt
done
