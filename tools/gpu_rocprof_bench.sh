#!/bin/bash
# rocprofv3 kernel statistics of `python bench.py <args>`; keeps only the small summaries under
# gpurun_out/$1 (the kernel trace itself is tens of MB and would blow gpurun's 64 MiB return cap).
# usage: tools/gpu_rocprof_bench.sh <tag> [bench.py args...]
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
work=/tmp/rocprof_$tag
rm -rf $work
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $work -o bench -- \
    python bench.py --no-cpu-baseline "$@" > $out/bench_under_rocprof.log 2>&1
f=$(find $work -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -80 "$f" > $out/kernel_stats_top80.csv
grep '^{' $out/bench_under_rocprof.log | tail -1 > $out/bench_under_rocprof.json
rm -rf $work
