#!/bin/bash
# bench line, same box, at the round's last kernels: one drawing at a time; 3 in flight with one-wave grids 128 / 192 / 256
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_last}; mkdir -p $O
run() { echo -n "$*: " | tee -a $O/summary.txt
  timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $O/last.json
  python -c "import json,sys; j=json.loads(open(sys.argv[1]).read()); c=j['config']; print('value %.4f' % j['value'], 'latency', (c.get('latency_s') or {}).get('mean'), {k: round(v,3) for k,v in c['stage_seconds_rank0'].items() if v is not None and not k.startswith('style_all')})" $O/last.json | tee -a $O/summary.txt; }
run --inflight 3 --onewave-grid 192
run --inflight 3 --onewave-grid 128
run --inflight 3 --onewave-grid 192
run --inflight 3 --onewave-grid 160
run --inflight 3 --onewave-grid 128
