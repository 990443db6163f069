#!/bin/bash
# geometry forward with the gathers of the next level requested ahead (4 / 5 active levels): tests + A/B
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3_37; mkdir -p $O; rm -f $O/ab.txt
timeout 400 python -m pytest tests/test_gpu_hashgrid.py -q -m gpu 2>&1 | grep -v Warn | tail -5 > $O/tests.txt; tail -2 $O/tests.txt
for i in 1 2; do
  for v in 1 0; do
    echo "DSU_FWD_AHEAD=$v $(DSU_FWD_AHEAD=$v timeout 200 python tools/nsr_stage_ab.py 1500 2>/dev/null | tail -1)" >> $O/ab.txt
  done
done
cat $O/ab.txt
