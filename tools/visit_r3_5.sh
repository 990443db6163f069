#!/bin/bash
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_mesh_post.py tests/test_gpu_nsr_native.py -q -m gpu 2>&1 | grep -v Warning | tail -60 > $O/tests.txt; tail -60 $O/tests.txt
