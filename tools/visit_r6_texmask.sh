#!/bin/bash
# texture backward with the forward's ReLU pattern (bf16 x 3 recompute of layer 1): tests, NSR stage A/B, kernel stats
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_texmask}; mkdir -p $O
V=drawingspinup_amd/variants
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_nsr_reference_step.py tests/test_gpu_nsr_native.py tests/test_gpu_nsr_step.py tests/test_gpu_nsr_model.py -q -x 2>&1 | grep -v Warn | tail -6 | tee -a $O/ab.txt
for rep in 1 2; do
  DSU_HIP_LIB=$V/libdsu_hip_texold.so timeout 300 python tools/nsr_stage_ab.py 3000 2>/dev/null | tail -1 | tee -a $O/ab.txt
  timeout 300 python tools/nsr_stage_ab.py 3000 2>/dev/null | tail -1 | tee -a $O/ab.txt
done
bash tools/nsr_stage_trace.sh ${1:-r6_texmask} 3000 | grep "texture\|total kernel" | tee -a $O/ab.txt
