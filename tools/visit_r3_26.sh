#!/bin/bash
# same-box A/B: k1 with hand-placed load waits (default) vs -DDSU_K1_ROLLED (compiler-placed, loop form)
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3_26; mkdir -p $O
for i in 1 2 3; do
  for v in default k1rolled; do
    if [ "$v" = default ]; then unset DSU_HIP_LIB; else export DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_$v.so; fi
    timeout 200 python tools/nsr_stage_ab.py 1500 2>$O/err.txt | tail -1 >> $O/ab.txt
  done
done
unset DSU_HIP_LIB
cat $O/ab.txt
