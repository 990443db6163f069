"""Shared body of train_stage1.py / train_stage2.py (3_style_translator/train_stage{1,2}.py).

The job description is the reference's configs/config_stage{1,2}.yaml (entry/config.py).
"""
import argparse
import os
import time

import torch

from ..style.training import ModelLogger, Trainer, build_model, build_optimizer

from . import config as C


def default_job(stage):
    """configs/config_stage<N>.yaml as shipped (entry/config.BUILTIN)."""
    import copy
    return copy.deepcopy(C.BUILTIN[f"config_stage{stage}"]["job"])


def run(stage, argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--uid", default="0dd66be9d0534b93a092d8c4c4dfd30a")
    ap.add_argument("--no_mask", action="store_true")
    ap.add_argument("--no_pos", action="store_true")
    if stage == 2:
        ap.add_argument("--no_edge", action="store_true")
    ap.add_argument("--config", default=None, help="a configs/config_stageN.yaml of the reference")
    ap.add_argument("--vgg", default=None, help="vgg19 state_dict (torchvision layout); default: "
                                                "DSU_VGG19_WEIGHTS or the torch hub cache")
    ap.add_argument("--random_vgg", action="store_true",
                    help="train against RANDOM vgg19 features (no ImageNet file available)")
    args = ap.parse_args(argv)

    # train_stage1.py:18-21: configs/config_stage<N>.yaml of the working directory (the reference
    # hard-codes the path), --config for another file, the shipped values when neither exists
    config = C.load_stage_job(stage, args.config)
    config["data_root"] = os.path.join(config["root_dir"], args.uid, "mesh", "blender_render")
    use_mask, use_pos = not args.no_mask, not args.no_pos
    use_edge = stage == 2 and not args.no_edge
    log_name = f"logs_stage{stage}"
    if use_mask:
        log_name += "_mask"
        config["generator"]["args"]["input_channels"] += 1
    if use_pos:
        log_name += "_pos"
        config["generator"]["args"]["input_channels"] += 2
    if use_edge:
        log_name += "_edge"
    log_folder = os.path.join(config["root_dir"], args.uid, "mesh", log_name)
    os.makedirs(log_folder, exist_ok=True)
    model_logger = ModelLogger(log_folder, torch.save)
    if args.config:
        model_logger.copy_file(args.config)

    device = config.get("device") or "cuda:0"
    generator = build_model(config["generator"]["type"], config["generator"]["args"], device)
    opt_generator = build_optimizer(config["opt_generator"]["type"], generator,
                                    config["opt_generator"]["args"])
    discriminator = build_model(config["discriminator"]["type"], config["discriminator"]["args"],
                                device)
    opt_discriminator = build_optimizer(config["opt_discriminator"]["type"], discriminator,
                                        config["opt_discriminator"]["args"])
    perc_args = dict(config["perception_loss"]["perception_model"]["args"])
    if args.vgg:
        perc_args["path"] = args.vgg
    if args.random_vgg:
        perc_args["random_init"] = True
    perception_loss_model = build_model(config["perception_loss"]["perception_model"]["type"],
                                        perc_args, device)

    trainer_config = dict(config["trainer"])
    trainer_config["testing_name_list"] = [f for f in os.listdir(config["data_root"])
                                           if not f.startswith(".")]
    trainer_config["post_dir"] = os.path.join(config["root_dir"], args.uid, "char")
    trainer = Trainer(data_root=config["data_root"], trainer_config=trainer_config,
                      opt_generator=opt_generator, opt_discriminator=opt_discriminator,
                      model_logger=model_logger, perception_loss_model=perception_loss_model,
                      perception_loss_weight=config["perception_loss"]["weight"],
                      use_mask=use_mask, use_pos=use_pos, use_edge=use_edge, device=device)
    start = time.time()
    trainer.train(generator, discriminator, int(config["trainer"]["epochs"]),
                  log_name.replace("logs", "res"), 0)
    print("Training finished, cost time: ", time.time() - start)
