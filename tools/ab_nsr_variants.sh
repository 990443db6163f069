#!/bin/bash
# A/B of staged kernel variants of the NSR step in ONE GPU visit.
#   here (no GPU):   bash tools/ab_nsr_variants.sh build
#   on the GPU box:  gpurun --timeout 500 -- 'bash tools/ab_nsr_variants.sh run'
# Output: gpurun_out/ab/<variant>.{tests,ab}.txt  (default library = variant "default").
# Staged for the next visit:
#   dinbatch  -DDSU_DIN_BATCH   k1: the 16 derivative factors of a hidden tile first, then its 16 dIn
#             MFMAs back to back (no VALU issue slots between MFMAs on the one accumulator)
#   texdin2   -DDSU_TEX_DIN_2ACC  texture backward: dIn into two alternating accumulators
#   l0int     -DDSU_L0_INTERLEAVED  k1: layer 0 with the two hidden tiles alternating per k-pair (pre-round-2 order)
#   din2acc   -DDSU_DIN_2ACC    k1: even / odd hidden units into two dIn accumulators (VALU between MFMAs
#             on different accumulators); summation order differs: check the gradient tests
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
VARIANTS="default dinbatch din2acc l0int texdin2"
case "${1:-}" in
  build)
    python -m drawingspinup_amd.build
    python -m drawingspinup_amd.build --variant dinbatch -DDSU_DIN_BATCH
    python -m drawingspinup_amd.build --variant din2acc -DDSU_DIN_2ACC
    python -m drawingspinup_amd.build --variant l0int -DDSU_L0_INTERLEAVED
    python -m drawingspinup_amd.build --variant texdin2 -DDSU_TEX_DIN_2ACC
    ;;
  run)
    mkdir -p gpurun_out/ab
    export PYTHONPATH=$R
    for v in $VARIANTS; do
      if [ "$v" = default ]; then unset DSU_HIP_LIB; else export DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_$v.so; fi
      timeout 150 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_nsr_step.py -q -m gpu 2>&1 | tail -3 > gpurun_out/ab/$v.tests.txt
      timeout 120 python tools/nsr_stage_ab.py 1500 2>/dev/null | tail -1 > gpurun_out/ab/$v.ab.txt
      echo "== $v"; tail -1 gpurun_out/ab/$v.tests.txt; cat gpurun_out/ab/$v.ab.txt
    done
    ;;
  *) echo "usage: $0 build|run"; exit 2;;
esac
