#!/bin/bash
# after the side stream went to normal priority: the bench line, the NSR / in-flight GPU tests, the driver's command
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_close2}; mkdir -p $O
timeout 600 python bench.py --steps 6 --warmup 1 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-200 $O/bench.json
timeout 400 python -m pytest tests/test_gpu_nsr_native.py tests/test_gpu_nsr_step.py tests/test_gpu_inflight.py tests/test_gpu_nsr_reference_step.py -q 2>&1 | grep -v Warn | tail -3 | tee $O/pytest_nsr_tail.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_driver_command.json; cut -c1-200 $O/bench_driver_command.json
