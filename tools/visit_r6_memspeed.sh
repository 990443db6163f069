#!/bin/bash
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_memspeed}; mkdir -p $O
timeout 300 python tools/mem_speed_probe.py 3 2>&1 | grep -v Warn | tee $O/mem_speed.txt
