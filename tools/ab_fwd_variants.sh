#!/bin/bash
# A/B of the geometry-forward variants (csrc/hashgrid.hip) in ONE GPU visit.
#   here (no GPU):   bash tools/ab_fwd_variants.sh build
#   on the GPU box:  gpurun --timeout 900 -- 'bash tools/ab_fwd_variants.sh run'
# Output: gpurun_out/abfwd/: default.tests.txt (hash-grid / NSR tests with the default library),
# probe_real.txt (tools/fwd_phase_probe.py on inputs captured from the optimisation), <variant>.ab.txt
# (tools/nsr_stage_ab.py: the NSR stage of the bench's data path).
#   pereval  -DDSU_FWD_PER_EVAL        the evaluation-by-evaluation kernel of rounds 1-4
#   w3       -DDSU_FWD_WAVES=3         register allocation for 3 waves per SIMD at every level count
#   jb4      -DDSU_FWD_JB=4            four hidden units per weight group in the MLP phase
#   pipe     -DDSU_FWD_PIPE=1          requests of level l + 1 ahead of the interpolation of level l
# Probe-only builds (wrong results by construction): -DDSU_FWD_ABLATE=1 (no MLP), =2 (no table traffic).
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
VARIANTS="${VARIANTS:-default pereval w3 jb4 pipe}"
case "${1:-}" in
  build)
    python -m drawingspinup_amd.build
    python -m drawingspinup_amd.build --variant pereval -DDSU_FWD_PER_EVAL
    python -m drawingspinup_amd.build --variant w3 -DDSU_FWD_WAVES=3
    python -m drawingspinup_amd.build --variant jb4 -DDSU_FWD_JB=4
    python -m drawingspinup_amd.build --variant pipe -DDSU_FWD_PIPE=1 "-DDSU_FWD_WAVES=(ACT<=4?3:2)"
    ;;
  run)
    mkdir -p gpurun_out/abfwd
    export PYTHONPATH=$R
    unset DSU_HIP_LIB
    timeout 300 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_nsr_step.py tests/test_gpu_nsr_reference_step.py tests/test_gpu_nsr_model.py -q -m gpu -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/abfwd/default.tests.txt | tail -4
    timeout 200 python tools/fwd_phase_probe.py capture /tmp/fwd_inputs.pt 2>/dev/null | tail -1 | tee gpurun_out/abfwd/probe_real.txt
    for v in $VARIANTS; do
      if [ "$v" = default ]; then unset DSU_HIP_LIB; else export DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_$v.so; fi
      timeout 100 python tools/fwd_phase_probe.py time /tmp/fwd_inputs.pt 2>/dev/null | tail -1 | tee -a gpurun_out/abfwd/probe_real.txt
      timeout 150 python tools/nsr_stage_ab.py ${STEPS:-3000} 2>/dev/null | tail -1 | tee gpurun_out/abfwd/$v.ab.txt
    done
    ;;
  *) echo "usage: $0 build|run"; exit 2;;
esac
