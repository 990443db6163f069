#!/bin/bash
# bench line, same box, after the round's kernel work: drawings in flight 3 / 4 / 2, one-wave grid 128 / 96
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_k_again}; mkdir -p $O
run() { echo -n "$*: " | tee -a $O/summary.txt
  timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); c=j['config']; print('value %.4f' % j['value'], 'latency %.2f' % c['latency_s']['mean'], {k: round(v,2) for k,v in c['stage_seconds_rank0'].items() if k in ('mv','nsr_fit','nsr_export','style')})" | tee -a $O/summary.txt; }
run --inflight 3
run --inflight 4
run --inflight 2
run --inflight 3 --onewave-grid 96
run --inflight 4 --onewave-grid 96
run --inflight 3
