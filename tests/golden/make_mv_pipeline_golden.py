"""Generate tests/golden/mv_pipeline_reference.npz by running the REFERENCE's own
MVDiffusionImagePipeline.__call__ (2_charactor_reconstructor/mvdiffusion/pipelines/
pipeline_mvdiffusion_image.py:299-508 with _encode_image :150-182, prepare_latents :254-269,
prepare_camera_embedding :271-296) UNMODIFIED on the CPU, driving the reference's own
UNetMV2DConditionModel (reduced width, float64), the way mv.py:64-86 calls it.

    python tests/golden/make_mv_pipeline_golden.py          # needs /root/reference

Real libraries where they are installed: Pillow and transformers' CLIPImageProcessor (the
`feature_extractor`).  Stand-ins (oracle/stubs, see its README): diffusers' DiffusionPipeline /
DDIMScheduler / VaeImageProcessor, torchvision's to_pil_image / to_tensor, xformers.  The CLIP
vision tower and the VAE are replaced by small fixed linear maps (they are tested on their own;
here they only have to make every input of the loop depend on the image route), so the fixture
pins the pipeline's GLUE: the 8-bit PIL detour of the f16 input batch, CLIP preprocessing, `* 2 - 1`
and `* scaling_factor`, sin|cos camera embedding, no classifier-free guidance at guidance_scale 1,
`cat([latents, image_latents], 1)`, the DDIM loop with eta = 1, `latents / scaling_factor`, decode,
denormalise.  Random draws (initial latents, per-step variance noise) are replaced by values that
are a pure function of their tag (oracle/mv_weights.det_tensor); the tests inject the same.
"""
import json
import os
import sys

import numpy as np
import torch

from transformers import CLIPImageProcessor  # the REAL one; imported before the stand-ins below
CLIPImageProcessor()                             # (transformers probes `torchvision` lazily)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/2_charactor_reconstructor"
sys.path.insert(0, os.path.join(ROOT, "oracle", "stubs"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from oracle import mv_weights  # noqa: E402
from oracle.mv_pipeline_aux import (LinearVAE, LinearClip, aux_state, input_image,  # noqa: E402
                                    camera_embeddings, det_noise)
from diffusers.schedulers import DDIMScheduler  # noqa: E402  (stub)
import diffusers.schedulers as stub_sched  # noqa: E402
from mvdiffusion.models.unet_mv2d_condition import UNetMV2DConditionModel  # noqa: E402
from mvdiffusion.pipelines import pipeline_mvdiffusion_image as ref_pipe  # noqa: E402

from make_mv_reference_golden import CFG as UNET_CFG  # noqa: E402  (same directory)

STEPS = 3
KEEP = [0, 5, 6, 11]


def main():
    torch.set_grad_enabled(False)
    cfg = dict(UNET_CFG, sample_size=32)
    unet = UNetMV2DConditionModel(**cfg).double().eval()
    names_shapes = [(k, tuple(v.shape)) for k, v in unet.state_dict().items()]
    unet.load_state_dict(mv_weights.synth_state_dict(names_shapes), strict=True)
    unet.enable_xformers_memory_efficient_attention()
    vae = aux_state(LinearVAE().double().eval(), "aux.vae.")
    clip = aux_state(LinearClip().double().eval(), "aux.clip.")

    draws = []

    def fake_randn(shape, generator=None, device=None, dtype=None, layout=None):
        draws.append(tuple(shape))
        return det_noise("draw.%d" % (len(draws) - 1), tuple(shape)).to(dtype)

    ref_pipe.randn_tensor = fake_randn            # prepare_latents (pipeline :266)
    stub_sched.randn_tensor = fake_randn          # DDIMScheduler.step variance noise

    pipe = ref_pipe.MVDiffusionImagePipeline(
        vae=vae, image_encoder=clip, unet=unet, scheduler=DDIMScheduler(), safety_checker=None,
        feature_extractor=CLIPImageProcessor(), requires_safety_checker=False, num_views=6)
    pipe.set_progress_bar_config(disable=True)

    rec = {}
    enc = pipe._encode_image

    def spy_encode(image_pil, *a, **k):
        rec["pil0"] = np.asarray(image_pil[0]).copy()
        e, l = enc(image_pil, *a, **k)
        rec["image_embeddings"], rec["image_latents"] = e.clone(), l.clone()
        return e, l
    pipe._encode_image = spy_encode
    cam_fn = pipe.prepare_camera_embedding

    def spy_cam(*a, **k):
        rec["camera"] = cam_fn(*a, **k).clone()
        return rec["camera"]
    pipe.prepare_camera_embedding = spy_cam
    steps = []

    img = input_image()
    imgs_in = img[None].expand(12, -1, -1, -1).contiguous()               # mv.py:70 (f16 batch)
    cam = camera_embeddings()
    out = pipe(imgs_in, cam, generator=None, output_type="pt", num_images_per_prompt=1,
               num_inference_steps=STEPS, guidance_scale=1.0, eta=1.0,       # pipe_validation_kwargs
               callback=lambda i, t, lat: steps.append((int(t), lat.clone()))).images
    assert out.shape == (12, 3, 256, 256) and len(steps) == STEPS and len(draws) == STEPS + 1
    print("timesteps", [t for t, _ in steps], "draws", draws)
    print("out rms", float(out.pow(2).mean().sqrt()), "min/max", float(out.min()), float(out.max()))
    arrays = {
        "steps": np.int64(STEPS), "timesteps": np.array([t for t, _ in steps]), "keep": np.array(KEEP),
        "pil0": rec["pil0"],
        "image_embeddings": rec["image_embeddings"].numpy().astype(np.float32),
        "image_latents": rec["image_latents"].numpy().astype(np.float32),
        "camera": rec["camera"].numpy().astype(np.float32),
        "out": out[KEEP].numpy().astype(np.float16),
        "cfg_json": np.array(json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()})),
        "names": np.array([n for n, _ in names_shapes]),
        "shapes": np.array([",".join(map(str, s)) for _, s in names_shapes]),
    }
    for i, (_, lat) in enumerate(steps):
        arrays["lat_%d" % (i + 1)] = lat.numpy().astype(np.float32)
    path = os.path.join(ROOT, "tests", "golden", "mv_pipeline_reference.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
