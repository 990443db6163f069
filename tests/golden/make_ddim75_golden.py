"""Generate tests/golden/ddim75_reference.npz: the latents of the full 75-step DDIM loop
(mvdiffusion/pipelines/pipeline_mvdiffusion_image.py:463-486, eta = 1, no guidance: mv.py:81) of
the float64 oracle (oracle/mv_ref.denoise_loop) on the SHIPPED UNet architecture at 16x16 latents,
with injected initial latents and per-step noise.  SURVEY.md 8(d): "after 75 steps with injected
noise: report, expect <= 2e-2".

    python tests/golden/make_ddim75_golden.py        # CPU only, no /root/reference needed; ~1-2 h
    python tests/golden/make_ddim75_golden.py 3 32   # first 3 steps at 32x32 latents (BASELINE
                                                     # configs[1]'s size) -> ddim3_lat32_reference.npz

The weights are not stored: they come from the seeded recipe `_init(UNetMV2DConditionModel(**FULL),
11).half()` of tests/test_gpu_unet.py, which the test repeats on the GPU box (CPU generator =
identical values).  Inputs likewise from the seeds of test_ddim_steps_vs_oracle_loop (31).  Stored:
the oracle's latents after steps 1, 2, 4, 8, 16, 32, 50, 75 (f16-rounded after every step like
the reference's f16 pipeline) and a checksum of the weights.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
KEEP = (1, 2, 4, 8, 16, 32, 50, 75)

if __name__ == "__main__":
    import test_gpu_unet as T
    from oracle import mv_ref as mr
    from drawingspinup_amd.mv.unet import UNetMV2DConditionModel
    from drawingspinup_amd.mv.pipeline import DEFAULT_CAMERA_EMBEDDING, MVDiffusionImagePipeline
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 75
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 16           # latent height = width
    torch.manual_seed(0)
    model = T._init(UNetMV2DConditionModel(**T.FULL), 11).half()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    checksum = float(sum(v.double().abs().sum() for v in sd.values()))
    g = torch.Generator().manual_seed(31)
    B = 12
    emb = (torch.randn(B, 1, 768, generator=g) * 0.5).half()
    img_lat = torch.randn(B, 4, S, S, generator=g).half()
    lat0 = torch.randn(B, 4, S, S, generator=g).half()
    noise = torch.randn(75, B, 4, S, S, generator=g).half()
    cam = MVDiffusionImagePipeline.prepare_camera_embedding(None, DEFAULT_CAMERA_EMBEDDING) \
        if False else None
    pipe = MVDiffusionImagePipeline(model, None, None)
    cam = pipe.prepare_camera_embedding(DEFAULT_CAMERA_EMBEDDING).cpu()
    del model
    t0 = time.time()
    ref = T._full_ref(sd)

    class Timed:
        n = 0

        def __call__(self, *a):
            out = ref(*a)
            Timed.n += 1
            print(f"step {Timed.n} done at {time.time() - t0:.0f} s", flush=True)
            return out
    lats = mr.denoise_loop(Timed(), lat0, img_lat, emb, cam, 75, noise, eta=1.0, run_steps=steps,
                           round_dtype=torch.float16)
    out = {"weights_abs_sum": np.float64(checksum), "steps": np.array([k for k in KEEP if k <= steps])}
    for k in KEEP:
        if k <= steps:
            out["lat_%d" % k] = lats[k - 1].float().numpy()
    name = "ddim75_reference.npz" if (steps, S) == (75, 16) else \
        ("ddim%d_lat%d_reference.npz" % (steps, S) if S != 16 else "ddim%d_probe.npz" % steps)
    out["latent_size"] = np.int64(S)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", list(out))
