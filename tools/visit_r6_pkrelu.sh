#!/bin/bash
# Softplus in packed-f32 pairs in the pipelined backward + ReLU / Softplus max as one v_max_f32 (default) vs the library
# forms (variant prev = -DDSU_PIPE_SP_SCALAR -DDSU_RELU_FMAXF): tests, per-kernel averages of the NSR stage (rocprofv3), same box
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_pkrelu}; mkdir -p $O
V=drawingspinup_amd/variants
timeout 900 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_nsr_reference_step.py tests/test_gpu_nsr_native.py tests/test_gpu_nsr_step.py tests/test_gpu_nsr_model.py -q 2>&1 | grep -v Warn | tail -3 | tee -a $O/summary.txt
for rep in 1 2; do
  DSU_HIP_LIB=$V/libdsu_hip_prev.so timeout 300 python tools/nsr_stage_ab.py 3000 2>/dev/null | tail -1 | tee -a $O/summary.txt
  timeout 300 python tools/nsr_stage_ab.py 3000 2>/dev/null | tail -1 | tee -a $O/summary.txt
done
echo "# prev" | tee -a $O/summary.txt
DSU_HIP_LIB=$V/libdsu_hip_prev.so bash tools/nsr_stage_trace.sh $(basename $O)/prev 3000 | grep "pipe\|texture\|scatter\|fwd_shared\|total kernel" | tee -a $O/summary.txt
echo "# default" | tee -a $O/summary.txt
bash tools/nsr_stage_trace.sh $(basename $O)/default 3000 | grep "pipe\|texture\|scatter\|fwd_shared\|total kernel" | tee -a $O/summary.txt
