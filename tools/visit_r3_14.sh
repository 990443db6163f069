#!/bin/bash
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_entry.py tests/test_gpu_shims.py -q -m gpu 2>&1 | grep -v Warning | tail -12
timeout 300 python tools/unet_forward_time.py 2>&1 | tail -5 | tee $O/unet_forward_time.txt
