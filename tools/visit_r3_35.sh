#!/bin/bash
# attention kernel with the next K / V tile prefetched into registers: tests + same-box A/B
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3_35; mkdir -p $O; rm -f $O/ab.txt
timeout 300 python -m pytest tests/test_gpu_attention.py -q -m gpu 2>&1 | grep -v Warn | tail -5 > $O/tests.txt; tail -2 $O/tests.txt
for i in 1 2; do
  for v in default attnold; do
    if [ "$v" = default ]; then unset DSU_HIP_LIB; else export DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_$v.so; fi
    echo "$v $(timeout 200 python tools/unet_forward_time.py 40 2>/dev/null | tail -1)" >> $O/ab.txt
  done
done
unset DSU_HIP_LIB
cat $O/ab.txt
