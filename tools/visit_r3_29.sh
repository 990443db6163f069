#!/bin/bash
# export-tail rows (remesh, thinning) on the GPU + the hash-grid suite after the NL=12 change
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3_29; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_thinning.py -q -m gpu 2>&1 | grep -v Warn | tail -40 > $O/thinning.txt; tail -40 $O/thinning.txt
timeout 300 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_mesh_post.py tests/test_gpu_mesh.py -q -m gpu -x 2>&1 | grep -v Warn | tail -4 > $O/other.txt; tail -3 $O/other.txt
