#!/bin/bash
# Round-6 memory-side (FETCH_SIZE / WRITE_SIZE) and SQ passes — run on an MI355X from the repository
# root: writes gpurun_out/<tag>/round6_pmc.json (copied to profiles/round6_pmc.json, which bench.py
# reads for `roofline.traffic` and the per-stage MFMA-busy shares).  PMC_WORKLOADS (default: all four)
# selects the workloads; every row is collected in this run (nothing carried over from earlier rounds).  Separate --pmc passes with
# --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; never
# combine --pmc with runtime / sys traces).  Workloads: tools/pmc_sdf_kernels.py (geometry network,
# N = 262 144 Morton-ordered samples, 5 levels), tools/pmc_unet_forward.py (one UNet forward, B = 12,
# 32x32 latents), tools/pmc_style_frame.py (stage 1 + stage 2 on one 512^2 frame).
tag=${1:-pmc6}
export TMPDIR=/tmp PYTHONPATH=$(pwd)
out=gpurun_out/$tag; mkdir -p $out
run_pass() {   # name counters... -- workload
  local name=$1; shift; export TEX_CHECK=0
  local counters=""
  while [ "$1" != "--" ]; do counters="$counters $1"; shift; done
  shift
  local w=/tmp/pmc6_${tag}_$name; rm -rf $w
  timeout 400 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $w -o p -- python "$@" \
      > $out/$name.log 2>&1
  local f=$(find $w -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" $out/$name.csv
  rm -rf $w
}
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
for wl in sdf:tools/pmc_sdf_kernels.py unet:tools/pmc_unet_forward.py style:tools/pmc_style_frame.py tex:tools/texture_time.py; do
  n=${wl%%:*}; s=${wl#*:}
  case " ${PMC_WORKLOADS:-sdf unet style tex} " in *" $n "*) ;; *) continue;; esac
  run_pass ${n}_fetch FETCH_SIZE -- $s
  run_pass ${n}_write WRITE_SIZE -- $s
  run_pass ${n}_sq $SQ -- $s
done
python - $out <<'P'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for path in sorted(glob.glob(os.path.join(out, "*.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        agg[name.split("(")[0][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            n = len(v)
            tail = v[n // 4:] or v                      # steady state: drop the first quarter
            res[k][c] = sum(tail) / len(tail)
            res[k]["dispatches"] = n
keep = ("sdf_fd", "reduce_partials", "conv_f16", "conv_igemm", "conv_x3", "mv_attention", "gemm_f16", "gn_", "groupnorm",
        "layernorm", "geglu", "deform", "style_conv", "texture_", "bin_")
js = {}
for k, cs in sorted(res.items()):
    if not any(s in k for s in keep):
        continue
    e = dict(cs)
    if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
        # KiB per dispatch; the guide's gfx950 correction: coalesced wide reads are reported at 1/2
        e["hbm_side_bytes"] = (2 * e.get("FETCH_SIZE", 0.0) + e.get("WRITE_SIZE", 0.0)) * 1024
    if e.get("SQ_BUSY_CYCLES"):
        e["mfma_busy_frac"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 4.0 / e["SQ_BUSY_CYCLES"] \
            if False else None
    if e.get("SQ_WAVE_CYCLES"):
        wc = e["SQ_WAVE_CYCLES"]
        e["active_frac"] = e.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        e["parked_frac"] = e.get("SQ_WAIT_ANY", 0.0) / wc
        e["issue_stall_frac"] = e.get("SQ_WAIT_INST_ANY", 0.0) / wc
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_WAVE_CYCLES quad-cycles per wave
        e["mfma_busy_of_wave_cycles"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 4.0 / wc
    e.pop("mfma_busy_frac", None)
    js[k] = e
json.dump({"note": "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE [KiB per dispatch], SQ_*), means over the "
                   "steady-state dispatches; hbm_side_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
                   "(MI355X_MICROARCH.md: gfx950 reports coalesced reads at 1/2)",
           "workloads": {"sdf": "tools/pmc_sdf_kernels.py: N = 262144 Morton-ordered samples, 5 active levels, "
                                "algorithmic bytes 315.6e6 per direction",
                         "unet": "tools/pmc_unet_forward.py", "style": "tools/pmc_style_frame.py",
                         "tex": "tools/texture_time.py (n = 262144 random samples)"},
           "sdf_algorithmic_bytes": 262144 * (7 * 5 * 8 * 4 + 84),
           "kernels": js}, open(os.path.join(out, "round6_pmc.json"), "w"), indent=1)
for k, e in js.items():
    print(f"{k[:60]:60s} " + " ".join(f"{c}={v:.4g}" for c, v in e.items() if isinstance(v, float)))
P
