#!/bin/bash
# bench line, same box: hardware queues available to the process's streams (3 drawings in flight = 6+ streams)
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_queues}; mkdir -p $O
for q in 4 8 16 4 8; do
  echo -n "GPU_MAX_HW_QUEUES=$q: " | tee -a $O/summary.txt
  GPU_MAX_HW_QUEUES=$q timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('value %.4f' % j['value'], 'latency', j['config']['latency_s'])" | tee -a $O/summary.txt
done
