#!/bin/bash
# A/B of the geometry-forward variants (csrc/hashgrid.hip) in ONE GPU visit.
#   here (no GPU):   bash tools/ab_fwd_variants.sh build
#   on the GPU box:  gpurun --timeout 900 -- 'bash tools/ab_fwd_variants.sh run'
# Output: gpurun_out/abfwd/<variant>.{tests,ab}.txt  (default library = variant "default").
#   pereval  -DDSU_FWD_PER_EVAL        the evaluation-by-evaluation kernel of rounds 1-4
#   tb2      -DDSU_FWD_TB=2            requests of one axis pair in flight instead of all six offsets
#   jb4      -DDSU_FWD_JB=4            four hidden units per weight group in the MLP phase
#   fence    -DDSU_FWD_LEVEL_FENCE=1   scheduling barrier between levels
#   w2       -DDSU_FWD_WAVES=2         register allocation for 2 waves per SIMD
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
VARIANTS="${VARIANTS:-default pereval tb2 jb4 fence w2}"
case "${1:-}" in
  build)
    python -m drawingspinup_amd.build
    python -m drawingspinup_amd.build --variant pereval -DDSU_FWD_PER_EVAL
    python -m drawingspinup_amd.build --variant tb2 -DDSU_FWD_TB=2
    python -m drawingspinup_amd.build --variant jb4 -DDSU_FWD_JB=4
    python -m drawingspinup_amd.build --variant fence -DDSU_FWD_LEVEL_FENCE=1
    python -m drawingspinup_amd.build --variant w2 -DDSU_FWD_WAVES=2
    ;;
  run)
    mkdir -p gpurun_out/abfwd
    export PYTHONPATH=$R
    for v in $VARIANTS; do
      if [ "$v" = default ]; then unset DSU_HIP_LIB; else export DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_$v.so; fi
      if [ "$v" = default ] || [ "$v" = pereval ]; then
        timeout 200 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_nsr_step.py tests/test_gpu_nsr_reference_step.py -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/abfwd/$v.tests.txt
        echo "== $v tests"; tail -3 gpurun_out/abfwd/$v.tests.txt
      fi
      timeout 150 python tools/nsr_stage_ab.py ${STEPS:-3000} 2>/dev/null | tail -1 > gpurun_out/abfwd/$v.ab.txt
      echo "== $v"; cat gpurun_out/abfwd/$v.ab.txt
    done
    ;;
  buildprobe)
    # builds for tools/fwd_phase_probe.py (the ablated ones give wrong results by construction)
    python -m drawingspinup_amd.build
    python -m drawingspinup_amd.build --variant pereval -DDSU_FWD_PER_EVAL
    python -m drawingspinup_amd.build --variant pipe -DDSU_FWD_PIPE=1 "-DDSU_FWD_WAVES=(ACT<=4?3:2)"
    python -m drawingspinup_amd.build --variant w3 "-DDSU_FWD_WAVES=(ACT<=4?3:2)"
    python -m drawingspinup_amd.build --variant s_nomlp -DDSU_FWD_ABLATE=1
    python -m drawingspinup_amd.build --variant s_noload -DDSU_FWD_ABLATE=2
    python -m drawingspinup_amd.build --variant p_nomlp -DDSU_FWD_PER_EVAL -DDSU_FWD_ABLATE=1
    python -m drawingspinup_amd.build --variant p_noload -DDSU_FWD_PER_EVAL -DDSU_FWD_ABLATE=2
    ;;
  probe)
    mkdir -p gpurun_out/abfwd
    export PYTHONPATH=$R
    unset DSU_HIP_LIB
    timeout 200 python tools/fwd_phase_probe.py capture /tmp/fwd_inputs.pt 2>/dev/null | tail -1 | tee gpurun_out/abfwd/probe_real.txt
    for v in default pereval pipe w3 s_nomlp s_noload p_nomlp p_noload; do
      if [ "$v" = default ]; then unset DSU_HIP_LIB; else export DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_$v.so; fi
      timeout 100 python tools/fwd_phase_probe.py time /tmp/fwd_inputs.pt 2>/dev/null | tail -1 | tee -a gpurun_out/abfwd/probe_real.txt
      timeout 100 python tools/fwd_phase_probe.py synthetic 2>/dev/null | tail -1 | tee -a gpurun_out/abfwd/probe_synthetic.txt
    done
    ;;
  *) echo "usage: $0 build|run|buildprobe|probe"; exit 2;;
esac
