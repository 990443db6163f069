#!/bin/bash
# last seconds of round 6: six short reconstructions, side stream high priority + pooled (one stream for all six drivers)
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/r6_pool_probe; mkdir -p $O
timeout 45 python tools/nsr_modes_probe.py 6 1000 1 1 2>$O/err.txt | grep '^{' | tee $O/high_pooled.txt
tail -3 $O/err.txt
