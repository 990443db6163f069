#!/bin/bash
# A/B of the staged kernel variants of the NSR step in ONE GPU visit.
#   here (no GPU):   bash tools/ab_nsr_variants.sh build
#   on the GPU box:  gpurun --timeout 400 -- 'bash tools/ab_nsr_variants.sh run'
# Output: gpurun_out/ab/<variant>.{tests,probe}.txt  (default library = variant "default").
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
case "${1:-}" in
  build)
    python -m drawingspinup_amd.build
    python -m drawingspinup_amd.build --variant noabl -DDSU_NO_ABLATE
    python -m drawingspinup_amd.build --variant noscan -DDSU_NO_ABLATE -DDSU_BWD_NOSCAN
    python -m drawingspinup_amd.build --variant occ4 -DDSU_FD_FWD_OCC4
    ;;
  run)
    mkdir -p gpurun_out/ab
    export PYTHONPATH=$R
    for v in default noabl noscan occ4; do
      if [ "$v" = default ]; then unset DSU_HIP_LIB; else export DSU_HIP_LIB=$R/drawingspinup_amd/variants/libdsu_hip_$v.so; fi
      timeout 120 python -m pytest tests/test_gpu_hashgrid.py tests/test_gpu_nsr_step.py -q -m gpu 2>&1 | tail -3 > gpurun_out/ab/$v.tests.txt
      timeout 120 python tools/nsr_train_probe.py 400 2>&1 | tail -3 > gpurun_out/ab/$v.probe.txt
      echo "== $v"; tail -1 gpurun_out/ab/$v.tests.txt; tail -2 gpurun_out/ab/$v.probe.txt
    done
    ;;
  *) echo "usage: $0 build|run"; exit 2;;
esac
