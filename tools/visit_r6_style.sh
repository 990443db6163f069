#!/bin/bash
# style / matting arithmetic visit: new exact-f32 packed kernels (tests, three-way timing, matting time, bench line)
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/${1:-r6_style}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_style.py tests/test_gpu_matting.py -q -x 2>&1 | grep -v Warning | tail -8 > $O/pytest_style_tail.txt; tail -4 $O/pytest_style_tail.txt
timeout 300 python tools/style_eval_time.py 4 5 > $O/style_eval_time.txt 2>&1; tail -3 $O/style_eval_time.txt
timeout 300 python tools/matting_time.py > $O/matting_time.txt 2>&1; tail -12 $O/matting_time.txt
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench.json; python - $O/bench.json <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(d["value"], d["config"]["stage_seconds_rank0"])
P
