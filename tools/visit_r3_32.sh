#!/bin/bash
# split-K target with the 64 x 64 tiles
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3_32; mkdir -p $O; rm -f $O/ab.txt
for t in 200 0 50 100 400; do
  echo "split_target=$t $(DSU_CONV_SPLIT_TARGET=$t timeout 200 python tools/unet_forward_time.py 40 2>/dev/null | tail -1)" >> $O/ab.txt
done
echo "fixup=1 target=200 $(DSU_SPLITK_FIXUP=1 timeout 200 python tools/unet_forward_time.py 40 2>/dev/null | tail -1)" >> $O/ab.txt
cat $O/ab.txt
