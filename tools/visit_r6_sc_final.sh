#!/bin/bash
# scatter workgroup shape adopted in round 6 (default) vs the shape of rounds 4-5 (scold): tests, bench line, same box
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_sc_final}; mkdir -p $O
V=drawingspinup_amd/variants
timeout 900 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_nsr_reference_step.py tests/test_gpu_nsr_native.py tests/test_gpu_nsr_step.py tests/test_gpu_nsr_model.py tests/test_gpu_shims.py -q 2>&1 | grep -v Warn | tail -3 | tee -a $O/summary.txt
for l in scold default scold default; do
  echo -n "bench $l: " | tee -a $O/summary.txt
  if [ $l = default ]; then timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/$l.json
  else DSU_HIP_LIB=$V/libdsu_hip_$l.so timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/$l.json; fi
  python -c "import json,sys; j=json.loads(open(sys.argv[1]).read()); c=j['config']; print('value %.4f' % j['value'], 'latency %.2f' % c['latency_s']['mean'], {k: round(v,2) for k,v in c['stage_seconds_rank0'].items() if k in ('mv','nsr_fit','nsr_export','style')}, 'pair alone %.4f ms' % j['roofline']['avg_launch_ms_alone'])" $O/$l.json | tee -a $O/summary.txt
done
