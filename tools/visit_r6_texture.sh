#!/bin/bash
# texture MLP backward, same box: rows one block ahead by LDS DMA (default) vs the form of rounds 2-5 (texold)
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_texture}; mkdir -p $O
V=drawingspinup_amd/variants
for rep in 1 2; do
  DSU_HIP_LIB=$V/libdsu_hip_texold.so timeout 200 python tools/texture_time.py 2>/dev/null | grep -v Warn | tee -a $O/texture_ab.txt
  timeout 200 python tools/texture_time.py 2>/dev/null | grep -v Warn | tee -a $O/texture_ab.txt
done
timeout 200 python tools/texture_time.py 70001 2>/dev/null | grep -v Warn | tee -a $O/texture_ab.txt
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_nsr_reference_step.py tests/test_gpu_nsr_native.py tests/test_gpu_nsr_step.py -q -x 2>&1 | grep -v Warn | tail -4 | tee -a $O/texture_ab.txt
for rep in 1 2; do
  for lib in $V/libdsu_hip_texold.so default; do
    if [ $lib = default ]; then timeout 300 python tools/nsr_stage_ab.py 3000 2>/dev/null | tail -1 | tee -a $O/texture_ab.txt
    else DSU_HIP_LIB=$lib timeout 300 python tools/nsr_stage_ab.py 3000 2>/dev/null | tail -1 | tee -a $O/texture_ab.txt; fi
  done
done
