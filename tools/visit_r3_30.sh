#!/bin/bash
# conv / GEMM kernel with loads two chunks ahead: tests, then same-box A/B of the UNet forward
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3_30; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_conv_f16.py tests/test_gpu_unet.py -q -m gpu -x 2>&1 | grep -v Warn | tail -5 > $O/tests.txt; tail -3 $O/tests.txt
for i in 1 2; do
  for v in default pf1; do
    if [ "$v" = default ]; then unset DSU_HIP_LIB; else export DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_$v.so; fi
    echo "$v $(timeout 200 python tools/unet_forward_time.py 40 2>/dev/null | tail -1)" >> $O/ab.txt
  done
done
unset DSU_HIP_LIB
cat $O/ab.txt
