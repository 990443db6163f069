#!/bin/bash
# round-3 visit 1: the 14 newly marked GEMM tests, A/B of the staged variants, SQ counters of the
# NSR step, BASELINE configs 3/4 bench lines.  Output under gpurun_out/r3v1/.
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v1; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_conv_f16.py -q -m gpu 2>&1 | tail -4 > $O/conv_tests.txt; cat $O/conv_tests.txt
for v in default dinbatch din2acc l0int texdin2; do
  if [ "$v" = default ]; then unset DSU_HIP_LIB; else export DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_$v.so; fi
  timeout 120 python tools/nsr_stage_ab.py 1500 2>/dev/null | tail -1 > $O/ab_$v.txt; echo "== $v"; cat $O/ab_$v.txt
done
unset DSU_HIP_LIB
timeout 150 python bench.py --config nsr50k --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_nsr50k.json; cat $O/bench_nsr50k.json
timeout 150 python bench.py --config frames --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_frames.json; cat $O/bench_frames.json
bash tools/pmc_sq_nsr.sh r3v1 300 2>&1 | tail -40
