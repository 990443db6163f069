#!/bin/bash
# bench line, same box, 3 drawings in flight: the round's bf16 x 3 conversions in the two one-wave-per-SIMD
# kernels (default) vs their f32 forms (allf32 variant)
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_bf16_bench}; mkdir -p $O
for l in allf32 default allf32 default; do
  echo -n "$l: " | tee -a $O/summary.txt
  if [ $l = default ]; then timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/$l.json
  else DSU_HIP_LIB=drawingspinup_amd/variants/libdsu_hip_$l.so timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/$l.json; fi
  python -c "import json,sys; j=json.loads(open(sys.argv[1]).read()); c=j['config']; print('value %.4f' % j['value'], 'latency %.2f' % c['latency_s']['mean'], {k: round(v,2) for k,v in c['stage_seconds_rank0'].items() if k in ('mv','nsr_fit','nsr_export','style')}, 'pair alone %.4f ms' % j['roofline']['avg_launch_ms_alone'])" $O/$l.json | tee -a $O/summary.txt
done
