"""Shared body of test_stage1.py / test_stage2.py (3_style_translator/test_stage{1,2}.py)."""
import argparse
import os
import time

import numpy as np
import torch
from PIL import Image

from ..style import build_model
from . import config as C
from . import data as D


def run(stage, argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--uid", default="0dd66be9d0534b93a092d8c4c4dfd30a")
    ap.add_argument("--no_mask", action="store_true")
    ap.add_argument("--no_pos", action="store_true")
    if stage == 2:
        ap.add_argument("--no_edge", action="store_true")
        ap.add_argument("--no_alpha", action="store_true")
    ap.add_argument("--checkpoint_id", type=int, default=99999)
    ap.add_argument("--root_dir", default=None, help="override of the job's root_dir")
    ap.add_argument("--random_init", action="store_true", help="no checkpoint: random weights")
    args = ap.parse_args(argv)
    # test_stage1.py:23-26: configs/config_stage<N>.yaml relative to the working directory (the
    # reference hard-codes the path); the shipped values when that file is not there
    config = C.load_stage_job(stage)
    if args.root_dir is not None:
        config["root_dir"] = args.root_dir
    args.root_dir = config["root_dir"]
    use_mask, use_pos = not args.no_mask, not args.no_pos
    use_edge = stage == 2 and not args.no_edge
    log_name = f"logs_stage{stage}" + ("_mask" if use_mask else "") + ("_pos" if use_pos else "") \
        + ("_edge" if use_edge else "")
    gen_args = dict(config["generator"]["args"])
    gen_args["input_channels"] += int(use_mask) + 2 * int(use_pos)
    pre_dir = config["trainer"]["pre_dir"]
    dev = torch.device(config.get("device") or "cuda:0")
    if dev.type != "cuda":
        raise RuntimeError("the style nets run on the HIP kernels: job.device must be a GPU")
    gen = build_model(config["generator"]["type"], gen_args, dev)
    ckpt = os.path.join(args.root_dir, args.uid, "mesh", log_name, "model_%05d.pth" % args.checkpoint_id)
    if not args.random_init:
        gen.load_state_dict(torch.load(ckpt, map_location=dev))
    gen.eval()
    data_root = os.path.join(args.root_dir, args.uid, "mesh/blender_render")
    result_folder = log_name.replace("logs", "res")
    start = time.time()
    for test_name in [f for f in os.listdir(data_root) if not f.startswith(".")]:
        root = os.path.join(data_root, test_name)
        ds = D.DatasetFullImages(root, pre_dir, use_mask, use_pos, use_edge)
        os.makedirs(os.path.join(root, result_folder), exist_ok=True)
        with torch.no_grad():
            for i in range(len(ds)):
                b = ds[i]
                out = gen(b["pre"][None].to(dev))[0]
                img = D.to_image_space(out.cpu().numpy()).transpose(1, 2, 0)
                if stage == 1 or not args.no_alpha:
                    alpha = (b["pre_mask"].numpy().transpose(1, 2, 0) * 255).astype(np.uint8)
                    img = np.concatenate((img, alpha), 2)
                Image.fromarray(img).save(os.path.join(root, result_folder, b["file_name"]))
    print(time.time() - start)
    print("Testing finished", flush=True)
