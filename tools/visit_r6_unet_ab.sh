#!/bin/bash
# UNet forward, same box: split-K publish (threadfence vs agent-scope release/acquire), split target, tile switch
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_unet_ab}; mkdir -p $O
V=drawingspinup_amd/variants
run() { echo -n "$1: " | tee -a $O/ab.txt; shift; env "$@" timeout 200 python tools/unet_forward_time.py 60 2>/dev/null | tail -1 | tee -a $O/ab.txt; }
for rep in 1 2; do
  run "threadfence (rounds 3-5)" DSU_HIP_LIB=$V/libdsu_hip_tf.so
  run "agent release/acquire" DSU_HIP_LIB=$V/libdsu_hip_ab.so
done
run "agent, split target 100" DSU_HIP_LIB=$V/libdsu_hip_ab.so DSU_CONV_SPLIT_TARGET=100
run "agent, split target 200" DSU_HIP_LIB=$V/libdsu_hip_ab.so DSU_CONV_SPLIT_TARGET=200
run "agent, split target 25" DSU_HIP_LIB=$V/libdsu_hip_ab.so DSU_CONV_SPLIT_TARGET=25
run "agent, small tiles below 256" DSU_HIP_LIB=$V/libdsu_hip_ab.so DSU_CONV_SMALL_MAX_TILES=256
run "agent, small tiles below 1024" DSU_HIP_LIB=$V/libdsu_hip_ab.so DSU_CONV_SMALL_MAX_TILES=1024
run "agent, small tiles below 128" DSU_HIP_LIB=$V/libdsu_hip_ab.so DSU_CONV_SMALL_MAX_TILES=128
timeout 300 python -m pytest tests/test_gpu_conv_f16.py tests/test_gpu_unet.py -q -x 2>&1 | grep -v Warn | tail -3 | tee -a $O/ab.txt
