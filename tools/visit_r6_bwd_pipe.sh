#!/bin/bash
# geometry backward, lean / pipelined MLP part: tests, same-box A/B of the NSR stage (variant build with the
# environment switches: DSU_BWD_PIPE=0 = general kernel), kernel statistics
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/${1:-r6_pipe}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_nsr_reference_step.py tests/test_gpu_nsr_native.py tests/test_gpu_nsr_step.py -q -x 2>&1 | grep -v Warning | tail -12 > $O/pytest_tail.txt; tail -5 $O/pytest_tail.txt
AB=drawingspinup_amd/variants/libdsu_hip_ab.so
for rep in 1 2; do
  for pipe in 0 1; do
    DSU_HIP_LIB=$AB DSU_BWD_PIPE=$pipe timeout 300 python tools/nsr_stage_ab.py 3000 2>/dev/null | tail -1 | sed "s/^/pipe=$pipe /" | tee -a $O/ab_nsr_stage.txt
  done
done
bash tools/nsr_stage_trace.sh ${1:-r6_pipe} 3000 | tail -16
