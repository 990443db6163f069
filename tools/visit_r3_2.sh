#!/bin/bash
# round-3 visit 2: new parity tests + the native NSR step driver (tests, A/B vs the Python-sequenced step)
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nsr_native.py tests/test_gpu_nsr_reference_step.py tests/test_gpu_mesh.py tests/test_gpu_nsr_step.py tests/test_gpu_nsr_model.py -q -m gpu 2>&1 | tail -40 > $O/tests_nsr.txt; cat $O/tests_nsr.txt
timeout 600 python -m pytest tests/test_gpu_unet.py -q -m gpu -k "ddim" -s 2>&1 | tail -15 > $O/tests_ddim.txt; cat $O/tests_ddim.txt
for mode in native fused; do
  DSU_STEP=$mode timeout 200 python tools/nsr_stage_ab.py 1500 2>&1 | tail -2 > $O/ab_$mode.txt; echo "== $mode"; cat $O/ab_$mode.txt
done
