#!/bin/bash
# NSR drawings back to back in one process (tools/nsr_modes_probe.py), side stream at high / normal / low priority (variant ab)
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_sideprio}; mkdir -p $O
for pr in 1 2 0 1 2; do
  echo "# DSU_NSR_SIDE_PRIO=$pr" | tee -a $O/summary.txt
  DSU_HIP_LIB=drawingspinup_amd/variants/libdsu_hip_ab.so DSU_NSR_SIDE_PRIO=$pr timeout 200 python tools/nsr_modes_probe.py 5 3000 2>/dev/null | grep '^{' | tee -a $O/summary.txt
done
