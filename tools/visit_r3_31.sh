#!/bin/bash
# 64 x 64 tile form of the f16 implicit-GEMM kernel: tests (default threshold and "always small"),
# then the UNet forward against the threshold
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3_31; mkdir -p $O; rm -f $O/ab.txt
timeout 300 python -m pytest tests/test_gpu_conv_f16.py -q -m gpu 2>&1 | grep -v Warn | tail -25 > $O/tests.txt; tail -3 $O/tests.txt
DSU_CONV_SMALL_MAX_TILES=100000 timeout 300 python -m pytest tests/test_gpu_conv_f16.py -q -m gpu 2>&1 | grep -v Warn | tail -25 > $O/tests_small.txt; tail -3 $O/tests_small.txt
for t in 0 512 100000; do
  echo "max_tiles=$t $(DSU_CONV_SMALL_MAX_TILES=$t timeout 200 python tools/unet_forward_time.py 40 2>/dev/null | tail -1)" >> $O/ab.txt
done
cat $O/ab.txt
