#!/bin/bash
# product library, priority through dsu_set_nsr_side_stream_priority: 5 short reconstructions at normal priority
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/r6_setter; mkdir -p $O
timeout 70 python tools/nsr_modes_probe.py 5 1000 2 2>/dev/null | grep '^{' | tee $O/normal.txt
