#!/bin/bash
# GroupNorm with 16-byte super-group accesses: norm tests on both paths, then same-box A/B of the UNet forward
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3_36; mkdir -p $O; rm -f $O/ab.txt
timeout 200 python -m pytest tests/test_gpu_unet.py -q -m gpu -k "norm_kernels or unet_small" 2>&1 | grep -v Warn | tail -12 > $O/tests.txt; tail -3 $O/tests.txt
DSU_GN_SUPER=0 timeout 200 python -m pytest tests/test_gpu_unet.py -q -m gpu -k "norm_kernels" 2>&1 | grep -v Warn | tail -5 > $O/tests_old.txt; tail -2 $O/tests_old.txt
for i in 1 2; do
  for v in 1 0; do
    echo "DSU_GN_SUPER=$v $(DSU_GN_SUPER=$v timeout 200 python tools/unet_forward_time.py 40 2>/dev/null | tail -1)" >> $O/ab.txt
  done
done
cat $O/ab.txt
