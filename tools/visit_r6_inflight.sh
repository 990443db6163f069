#!/bin/bash
# bench line, same box: drawings in flight per GPU, start skew, NSR slots
set -u
export PYTHONPATH=$(pwd) TMPDIR=/tmp
O=gpurun_out/${1:-r6_inflight}; mkdir -p $O
run() {
  tag=$1; shift
  timeout 900 python bench.py --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline "$@" 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python - $O/bench_$tag.json "$tag $*" <<'P'
import json, sys
j = json.loads(open(sys.argv[1]).read())
c = j["config"]
print(sys.argv[2], "| value %.4f" % j["value"], "ms_per_step %.0f" % j["ms_per_step"], "latency", {k: round(v, 2) for k, v in (c.get("latency_s") or {}).items()},
      "stages", {k: round(v, 2) for k, v in c["stage_seconds_rank0"].items() if v is not None and k in ("mv", "nsr_fit", "nsr_export", "style")})
P
}
{
run k1 --inflight 1
run k2s --inflight 2
run k3s --inflight 3
run k2g1 --inflight 2 --nsr-slots 1
run k3g1 --inflight 3 --nsr-slots 1
run k3g2 --inflight 3 --nsr-slots 2
run k3s0 --inflight 3 --inflight-skew 0
} 2>&1 | tee $O/summary2.txt
