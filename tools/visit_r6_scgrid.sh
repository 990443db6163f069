#!/bin/bash
# bench line, same box, 3 drawings in flight: workgroups of the scatter kernel (LDS-exclusive: 150 KB each)
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_scgrid}; mkdir -p $O
for g in 256 128 192 96 256 128; do
  echo -n "scatter grid $g: " | tee -a $O/summary.txt
  timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --scatter-grid $g 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); c=j['config']; print('value %.4f' % j['value'], 'latency %.2f' % c['latency_s']['mean'], {k: round(v,2) for k,v in c['stage_seconds_rank0'].items() if k in ('mv','nsr_fit','nsr_export','style','contour')})" | tee -a $O/summary.txt
done
