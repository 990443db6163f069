"""Time the REFERENCE's training loop body (trainers.py:148-172) on the host CPU at the shipped
stage-2 configuration (GeneratorJ: nn.Conv2d only, so nothing third-party is shimmed in the
timed path except the seeded VGG19 stack).  Needs /root/reference; informational baseline for
DESIGN.md, not part of bench.py.

    python tools/style_train_cpu_reference.py [--iters 2]
"""
import argparse
import importlib.util
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location(
    "make_style_train_golden", os.path.join(ROOT, "tests", "golden", "make_style_train_golden.py"))
g = importlib.util.module_from_spec(spec)
spec.loader.exec_module(g)          # installs the shims and imports the reference modules


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--stage", type=int, default=2)
    a = ap.parse_args()
    torch.manual_seed(0)
    args = dict(use_bias=False, tanh=True, append_smoothers=True, resnet_blocks=7,
                filters=[32, 64, 128, 128, 128, 64], input_channels=6)
    gen = getattr(g.ref_models, "GeneratorJ" if a.stage == 2 else "GeneratorJ_RIC")(**args)
    disc = g.ref_models.DiscriminatorN_IN(num_filters=12, n_layers=2)
    perc = g.ref_models.PerceptualVGG19(feature_layers=[0, 3, 5], use_normalization=False)
    opt_g = g.ref_trainers.build_optimizer("Adam", gen, dict(g.OPT))
    opt_d = g.ref_trainers.build_optimizer("Adam", disc, dict(g.OPT))
    t = g.make_trainer(perc)
    B, P = 40, 32
    times = []
    for it in range(a.iters + 1):
        batch = {"pre": torch.rand(B, 6, P, P) * 2 - 1, "pre_mask": (torch.rand(B, 1, P, P) > 0.3).float(),
                 "post": torch.rand(B, 3, P, P) * 2 - 1, "already": torch.rand(B, 3, P, P) * 2 - 1,
                 "already_mask": (torch.rand(B, 1, P, P) > 0.3).float()}
        t0 = time.time()
        gen.train(); disc.train()
        opt_d.zero_grad()
        d_loss = t.compute_discriminator_loss(gen, disc, batch)
        d_loss.backward()
        opt_d.step()
        opt_g.zero_grad()
        li, lp, la, _ = t.compute_generator_loss(gen, disc, batch, use_gan=True, use_mask=False)
        (t.reconstruction_weight * li + t.perception_loss_weight * lp + t.adversarial_weight * la).backward()
        opt_g.step()
        times.append(time.time() - t0)
    print({"stage": a.stage, "threads": torch.get_num_threads(), "s_per_iter": times[1:],
           "first_iter_s": times[0]})


if __name__ == "__main__":
    main()
