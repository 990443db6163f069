#!/bin/bash
# scatter kernel as 256-thread workgroups with 65 KB of LDS, two per CU (variant sc256) vs 512 threads / 150 KB / one per CU:
# NSR stage alone and the bench line with 3 drawings in flight, same box
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_sc256}; mkdir -p $O
V=drawingspinup_amd/variants
timeout 600 env DSU_HIP_LIB=$V/libdsu_hip_sc256.so python -m pytest tests/test_gpu_hashgrid.py -q -x 2>&1 | tail -1 | tee -a $O/summary.txt
for rep in 1 2; do
  DSU_HIP_LIB=$V/libdsu_hip_sc256.so timeout 300 python tools/nsr_stage_ab.py 3000 2>/dev/null | tail -1 | tee -a $O/summary.txt
  timeout 300 python tools/nsr_stage_ab.py 3000 2>/dev/null | tail -1 | tee -a $O/summary.txt
done
for l in sc256 default sc256 default; do
  echo -n "bench $l: " | tee -a $O/summary.txt
  if [ $l = default ]; then timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/$l.json
  else DSU_HIP_LIB=$V/libdsu_hip_$l.so timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/$l.json; fi
  python -c "import json,sys; j=json.loads(open(sys.argv[1]).read()); c=j['config']; print('value %.4f' % j['value'], 'latency %.2f' % c['latency_s']['mean'], {k: round(v,2) for k,v in c['stage_seconds_rank0'].items() if k in ('mv','nsr_fit','nsr_export','style')})" $O/$l.json | tee -a $O/summary.txt
done
