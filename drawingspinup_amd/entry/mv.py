"""`python mv.py --uid U [--all]` of the reference (2_charactor_reconstructor/mv.py:161-181)."""
import argparse
import json
import os

import torch
from PIL import Image

from .. import dist as ddist
from ..mv.pipeline import build_random_pipeline
from . import data as D


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--uid", default="0dd66be9d0534b93a092d8c4c4dfd30a")
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--img_fn", default="char/ffc_resnet_inpainted.png")
    ap.add_argument("--save_folder", default="mv")
    ap.add_argument("--data_root", default="../dataset/AnimatedDrawings/preprocessed")
    ap.add_argument("--uid_list_file", default="../dataset/AnimatedDrawings/drawings_uids.json")
    ap.add_argument("--pose_dir", default=None, help=".../mvdiffusion/data/fixed_poses/nine_views")
    ap.add_argument("--unet_state_dict", default=None, help="UNet state_dict (diffusers key names)")
    ap.add_argument("--seed", type=int, default=123456)         # configs/mvdiffusion-joint-ortho-6views.yaml:1
    ap.add_argument("--num_inference_steps", type=int, default=75)
    args = ap.parse_args(argv)
    rank, world, local = ddist.init()
    dev = torch.device("cuda", local)
    pipe = build_random_pipeline(dev, seed=0)
    if args.unet_state_dict:
        pipe.unet.load_state_dict(torch.load(args.unet_state_dict, map_location="cpu"))
    for m in (pipe.unet, pipe.vae, pipe.image_encoder):
        ddist.broadcast_module(m, 0)
    uids = json.load(open(args.uid_list_file)) if args.all else [args.uid]
    for uid in ddist.shard(uids, rank, world):
        img_fn = os.path.join(args.data_root, uid, args.img_fn)
        if not os.path.exists(img_fn):
            img_fn = os.path.join(args.data_root, uid, "char/texture.png")
        single = Image.open(img_fn).convert("RGBA")
        imgs, cam = D.mv_batch(single, args.pose_dir)
        g = torch.Generator(device=dev).manual_seed(args.seed)
        out = pipe(imgs.to(dev), cam.to(dev), generator=g, guidance_scale=1.0, output_type="pt",
                   eta=1.0, num_inference_steps=args.num_inference_steps)
        D.write_mv_outputs(os.path.join(args.data_root, uid, args.save_folder), out[:6], out[6:], single)
        print(uid, flush=True)


if __name__ == "__main__":
    main()
