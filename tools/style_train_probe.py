"""Time the per-character training iteration of the style translator at the
shipped configuration: batch 40 x 32x32 patches, GeneratorJ_RIC (stage 1) / GeneratorJ
(stage 2) filters [32,64,128,128,128,64] x 7 resnet blocks, DiscriminatorN_IN(12, 2 layers),
PerceptualVGG19([0,3,5]), Adam.  Synthetic 512x512 rest-pose render.

    python tools/style_train_probe.py [--stage 1|2] [--iters 30] [--warmup 5]
"""
import argparse
import json
import time

import numpy as np
import torch
from PIL import Image

from drawingspinup_amd.entry._train_stage import default_job
from drawingspinup_amd.style import training as T


def synthetic_dataset(dev, size=512, seed=0):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:size, 0:size]
    alpha = (((yy - size / 2) ** 2 / (0.42 * size) ** 2 + (xx - size / 2) ** 2 / (0.25 * size) ** 2)
             < 1).astype(np.uint8) * 255
    def rgb():
        return rng.randint(0, 256, (size, size, 3)).astype(np.uint8)
    pre = Image.fromarray(np.dstack([rgb(), alpha]))
    pos = Image.fromarray(np.dstack([rgb(), alpha]))
    post = Image.fromarray(rgb())
    return T.DatasetPatches_M.from_images(pre, post, Image.fromarray(alpha), pos, 32, True, True, dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", type=int, default=1)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    job = default_job(a.stage)
    job["generator"]["args"]["input_channels"] += 3
    gen = T.build_model(job["generator"]["type"], job["generator"]["args"], dev)
    disc = T.build_model("DiscriminatorN_IN", job["discriminator"]["args"], dev)
    perc = T.build_model("PerceptualVGG19", dict(job["perception_loss"]["perception_model"]["args"], random_init=True), dev)
    ds = synthetic_dataset(dev)
    tr = T.Trainer(None, dict(job["trainer"]), T.build_optimizer("Adam", disc, job["opt_discriminator"]["args"]),
                   T.build_optimizer("Adam", gen, job["opt_generator"]["args"]), None, perc,
                   job["perception_loss"]["weight"], True, True, False, dev, dataset=ds)
    tr.use_adversarial_loss = True
    np.random.seed(0)
    bs = job["trainer"]["batch_size"]
    for _ in range(a.warmup):
        tr.train_step(gen, disc, ds.batch(bs))
    torch.cuda.synchronize()
    t0 = time.time()
    t_data = 0.0
    for _ in range(a.iters):
        td = time.time()
        batch = ds.batch(bs)
        t_data += time.time() - td
        log = tr.train_step(gen, disc, batch)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.iters
    iters_total = (len(ds) // bs) * job["trainer"]["epochs"]
    print(json.dumps({"stage": a.stage, "ms_per_iter": dt * 1e3, "host_batch_ms": t_data / a.iters * 1e3,
                      "valid_pixels": len(ds), "iters_per_training": iters_total,
                      "training_s_extrapolated": dt * iters_total,
                      "losses": {k: float(v) for k, v in log.items()}}))


if __name__ == "__main__":
    main()
