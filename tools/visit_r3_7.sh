#!/bin/bash
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_entry.py -q -m gpu -k "vae or entry" 2>&1 | grep -v Warning | tail -30 > $O/tests.txt; tail -12 $O/tests.txt
timeout 300 python tools/unet_op_times.py 2>&1 | tail -60 > $O/unet_op_times.txt; cat $O/unet_op_times.txt
