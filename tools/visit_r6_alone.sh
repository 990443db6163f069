#!/bin/bash
# the two modes of roofline.alone: three processes of the bench (3 in flight, 2 steps), each repeating the drawing alone
# (DSU_ALONE_REPEAT=2: twice more, allocator cache dropped before the last); then one drawing at a time at the last kernels
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_alone}; mkdir -p $O
for p in 1 2 3; do
  echo "# process $p" | tee -a $O/summary.txt
  DSU_ALONE_REPEAT=2 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>$O/err$p.txt | tail -1 > $O/b$p.json
  grep "^\[alone" $O/err$p.txt | tee -a $O/summary.txt
  python -c "import json,sys; j=json.loads(open(sys.argv[1]).read()); print('value %.4f' % j['value'])" $O/b$p.json | tee -a $O/summary.txt
done
timeout 600 python bench.py --inflight 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/one_at_a_time.json
python -c "import json,sys; j=json.loads(open(sys.argv[1]).read()); c=j['config']; print('one at a time: value %.4f' % j['value'], {k: round(v,3) for k,v in c['stage_seconds_rank0'].items() if v is not None and not k.startswith('style_all')})" $O/one_at_a_time.json | tee -a $O/summary.txt
