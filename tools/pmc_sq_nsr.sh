#!/bin/bash
# SQ-counter passes over the NSR stage (tools/nsr_stage_ab.py, 300 steps) — run on an MI355X from
# the repository root; writes gpurun_out/<tag>/sq_<pass>.txt (per kernel: mean counter values of the
# steady-state dispatches).  Two passes of <= 8 SQ counters (MI355X_MICROARCH.md: 8 SQ slots per
# pass; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES
# cycles).  Reading (fractions of SQ_WAVE_CYCLES): active = SQ_ACTIVE_INST_ANY, parked on
# s_waitcnt / barriers = SQ_WAIT_ANY, issue stalls = SQ_WAIT_INST_ANY (LDS share: SQ_WAIT_INST_LDS),
# matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 4.  Never combine --pmc with runtime / sys traces.
# usage: tools/pmc_sq_nsr.sh <tag> [steps]
tag=${1:-sq}; steps=${2:-300}
export TMPDIR=/tmp PYTHONPATH=$(pwd)
out=gpurun_out/$tag; mkdir -p $out
pass_a="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
pass_b="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
i=0
for counters in "$pass_a" "$pass_b"; do
  i=$((i + 1)); w=/tmp/sq_${tag}_$i; rm -rf $w
  timeout 300 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $w -o sq -- \
      python tools/nsr_stage_ab.py $steps > $out/sq_pass$i.log 2>&1
  f=$(find $w -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" > $out/sq_pass$i.txt <<'P'
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
keep = ("sdf_fd", "texture_", "ray_march", "composite", "ray_losses", "bin_", "table_adamw")
for k, cs in sorted(agg.items()):
    if not any(s in k for s in keep):
        continue
    n = len(next(iter(cs.values())))
    print(f"{k:50s} n={n:5d} " + "  ".join(f"{c}={sum(v[n // 4:]) / max(len(v[n // 4:]), 1):.4g}" for c, v in sorted(cs.items())))
P
  rm -rf $w
  cat $out/sq_pass$i.txt
done
