#!/bin/bash
# geometry forward with the offset evaluations in packed-f32 pairs (default) vs one evaluation per instruction (variant
# fwdpk0 = -DDSU_FWD_PK=0): bit-identity tests, NSR stage alone, bench line with 3 drawings in flight — same box
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_fwdpk}; mkdir -p $O
V=drawingspinup_amd/variants
timeout 900 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_nsr_reference_step.py tests/test_gpu_nsr_native.py tests/test_gpu_nsr_step.py tests/test_gpu_nsr_model.py -q 2>&1 | grep -v Warn | tail -3 | tee -a $O/summary.txt
for rep in 1 2; do
  DSU_HIP_LIB=$V/libdsu_hip_fwdpk0.so timeout 300 python tools/nsr_stage_ab.py 3000 2>/dev/null | tail -1 | tee -a $O/summary.txt
  timeout 300 python tools/nsr_stage_ab.py 3000 2>/dev/null | tail -1 | tee -a $O/summary.txt
done
for l in fwdpk0 default fwdpk0 default; do
  echo -n "bench $l: " | tee -a $O/summary.txt
  if [ $l = default ]; then timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/$l.json
  else DSU_HIP_LIB=$V/libdsu_hip_$l.so timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/$l.json; fi
  python -c "import json,sys; j=json.loads(open(sys.argv[1]).read()); c=j['config']; print('value %.4f' % j['value'], 'latency %.2f' % c['latency_s']['mean'], {k: round(v,2) for k,v in c['stage_seconds_rank0'].items() if k in ('mv','nsr_fit','nsr_export','style')})" $O/$l.json | tee -a $O/summary.txt
done
