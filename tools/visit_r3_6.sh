#!/bin/bash
set -u
R=$(pwd); export PYTHONPATH=$R TMPDIR=/tmp
O=gpurun_out/r3v6; mkdir -p $O
timeout 900 python -m pytest tests/test_contour_host.py tests/test_gpu_unet.py tests/test_gpu_entry.py tests/test_gpu_style.py -q -m gpu -k "not ddim_75" 2>&1 | grep -v Warning | tail -40 > $O/tests.txt; tail -30 $O/tests.txt
python - <<'P' 2>&1 | tail -6 | tee $O/contour_time.txt
import time, torch, os
from drawingspinup_amd.contour.ffc import LAMA_FOURIER_GENERATOR, make_generator
dev = torch.device("cuda:0")
torch.manual_seed(0)
gen = make_generator(**LAMA_FOURIER_GENERATOR).eval().to(dev)
x = torch.rand(1, 4, 512, 512, device=dev)
for mode in ("hip", "torch"):
    os.environ["DSU_CONTOUR"] = mode
    with torch.no_grad():
        t0 = time.time(); y = gen(x); torch.cuda.synchronize(); first = time.time() - t0
        t0 = time.time()
        for _ in range(10): y = gen(x)
        torch.cuda.synchronize()
    print(mode, "first %.3f s, steady %.2f ms" % (first, (time.time() - t0) * 100), float(y.mean()))
    ref = y if mode == "hip" else ref
    if mode == "torch": print("max |hip - torch| =", float((ref - y).abs().max()))
P
