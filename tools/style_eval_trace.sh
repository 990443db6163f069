#!/bin/bash
# rocprofv3 kernel statistics of the two generators' evaluation (tools/style_eval_time.py);
# keeps the summary under gpurun_out/$1.   usage: tools/style_eval_trace.sh <tag> [batch] [reps]
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
work=/tmp/rocprof_$tag
rm -rf $work
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $work -o st -- \
    python $GRAFT_REPO_ROOT/tools/style_eval_time.py "$@" > $GRAFT_REPO_ROOT/$out/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $work -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -40 "$f" > $out/kernel_stats_top40.csv
rm -rf $work
