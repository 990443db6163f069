#!/bin/bash
# UNet forward, same box: split-K finished by a separate reduce launch (product) vs in the kernel by the last
# workgroup of a tile, with the two publish forms (threadfence per thread / one agent-scope release + acquire)
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_unet_ab2}; mkdir -p $O
V=drawingspinup_amd/variants
run() { echo -n "$1: " | tee -a $O/ab.txt; shift; env "$@" timeout 200 python tools/unet_forward_time.py 60 2>/dev/null | tail -1 | tee -a $O/ab.txt; }
for rep in 1 2; do
  run "reduce launch (product)" DSU_SPLITK_FIXUP=0
  run "in-kernel, agent release/acquire" DSU_SPLITK_FIXUP=1
  run "in-kernel, threadfence" DSU_SPLITK_FIXUP=1 DSU_HIP_LIB=$V/libdsu_hip_tf.so
done
DSU_SPLITK_FIXUP=1 timeout 600 python -m pytest tests/test_gpu_conv_f16.py tests/test_gpu_unet.py -q -x 2>&1 | grep -v Warn | tail -3 | tee -a $O/ab.txt
