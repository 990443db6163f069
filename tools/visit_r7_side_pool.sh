#!/bin/bash
# FIRST VISIT OF THE NEXT ROUND (prepared at the end of round 6, nothing of it measured yet): "high priority without the slow mode".
#  1) six reconstructions back to back, side stream high + pooled: does the fourth stay fast?  (un-pooled high: slow from the 4th on)
#  2) the bench line with three drawings in flight, same box, alternating: high (default) / high + pool / normal / normal + pool
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r7_side_pool}; mkdir -p $O
echo "# high, pooled" | tee -a $O/summary.txt
timeout 200 python tools/nsr_modes_probe.py 6 3000 1 1 2>/dev/null | grep '^{' | tee -a $O/summary.txt
echo "# high, one stream per driver" | tee -a $O/summary.txt
timeout 200 python tools/nsr_modes_probe.py 6 3000 1 0 2>/dev/null | grep '^{' | tee -a $O/summary.txt
run() { echo -n "$*: " | tee -a $O/summary.txt
  timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $O/last.json
  python -c "import json,sys; j=json.loads(open(sys.argv[1]).read()); c=j['config']; print('value %.4f' % j['value'], {k: round(v,2) for k,v in c['stage_seconds_rank0'].items() if k in ('mv','nsr_fit','nsr_export','style')}, [(r['kernel'], round(r['avg_launch_ms'],4)) for r in j['roofline']['alone'][:2]])" $O/last.json | tee -a $O/summary.txt; }
for rep in 1 2; do
  run --side-priority 1 --side-pool 0
  run --side-priority 1 --side-pool 1
  run --side-priority 2 --side-pool 0
  run --side-priority 2 --side-pool 1
done
