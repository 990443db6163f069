#!/bin/bash
# parity-chain visit: native-step / refresh / matting / entry tests, then a bench line
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/${1:-r6_parity}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_nsr_native.py tests/test_gpu_matting.py tests/test_gpu_entry.py tests/test_gpu_hashgrid.py -q -x 2>&1 | grep -v Warning | tail -25 > $O/pytest_tail.txt; tail -12 $O/pytest_tail.txt
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench.json; python - $O/bench.json <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(d["value"], d["config"]["stage_seconds_rank0"])
P
tail -5 $O/bench.err
