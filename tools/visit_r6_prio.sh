#!/bin/bash
# bench line, same box, 3 drawings in flight: NSR optimisation on a low-priority stream of its own
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_prio}; mkdir -p $O
for p in 0 1 0 1; do
  echo -n "fit-priority $p: " | tee -a $O/summary.txt
  timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --fit-priority $p 2>$O/err_$p.txt | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); c=j['config']; print('value %.4f' % j['value'], 'latency %.2f' % c['latency_s']['mean'], {k: round(v,2) for k,v in c['stage_seconds_rank0'].items() if k in ('mv','nsr_fit','nsr_export','style','contour')})" | tee -a $O/summary.txt
done
tail -3 $O/err_1.txt
