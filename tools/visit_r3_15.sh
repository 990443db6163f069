#!/bin/bash
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_f16.py tests/test_gpu_unet.py -q -m gpu -k "not ddim_75" 2>&1 | grep -v Warning | tail -12
timeout 300 python tools/unet_forward_time.py 2>&1 | tail -2 | tee $O/unet_forward_time.txt
DSU_SPLITK_FIXUP=0 timeout 300 python tools/unet_forward_time.py 2>&1 | tail -1 | tee -a $O/unet_forward_time.txt
