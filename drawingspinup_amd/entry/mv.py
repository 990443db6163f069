"""`python mv.py [--config configs/mvdiffusion-joint-ortho-6views.yaml] --uid U [--all]` of the
reference (2_charactor_reconstructor/mv.py:161-181).  The YAML file (loaded as mv.py:21-26 does,
entry/config.py) supplies seed, data_root, uid_list_file, views, resolution, the dataset's
img_wh and the pipeline's eta / guidance_scale / num_inference_steps; the remaining flags are
overrides for runs outside a reference checkout (None = the config's value).  The checkpoint is
`pretrained_model_name_or_path` when it names a local diffusers-layout directory ('./ckpts',
yaml:2) — the hub id the reference downloads is not reachable from here."""
import argparse
import json
import os

import torch
from PIL import Image

from .. import dist as ddist
from ..mv.pipeline import build_random_pipeline
from . import data as D


def main(argv=None):
    ap = argparse.ArgumentParser(description="mv generation")
    ap.add_argument("--config", default="./configs/mvdiffusion-joint-ortho-6views.yaml")   # mv.py:164
    ap.add_argument("--uid", default="0dd66be9d0534b93a092d8c4c4dfd30a")
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--img_fn", default="char/ffc_resnet_inpainted.png")
    ap.add_argument("--save_folder", default="mv")
    ap.add_argument("--data_root", default=None)
    ap.add_argument("--uid_list_file", default=None)
    ap.add_argument("--pose_dir", default=None, help=".../mvdiffusion/data/fixed_poses/nine_views")
    ap.add_argument("--pretrained", default=None,
                    help="local diffusers-layout directory of the Wonder3D checkpoint "
                         "(unet/, vae/, image_encoder/): what mv.py:29-39 downloads from the hub")
    ap.add_argument("--unet_state_dict", default=None, help="UNet weights file (diffusers key names)")
    ap.add_argument("--vae_state_dict", default=None, help="VAE weights file (diffusers key names)")
    ap.add_argument("--image_encoder", default=None, help="local CLIP vision directory (config + weights)")
    ap.add_argument("--random_init", action="store_true",
                    help="run with random weights for whatever was not given (smoke tests only: the "
                         "outputs are noise)")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--num_inference_steps", type=int, default=None)
    ap.add_argument("--matting", default="silhouette", choices=["silhouette", "isnet"],
                    help="side-view masks (mv.py:113-122): 'isnet' = remove_background through the "
                         "IS-Net session of mv/matting.py (needs --isnet_weights to mean anything), "
                         "'silhouette' = the filled not-white silhouette stand-in")
    ap.add_argument("--isnet_weights", default=None,
                    help="DIS IS-Net state_dict (isnet-general-use.pth); mv.py:17 loads "
                         "dis_pretrained/isnet_dis.onnx, the same network exported to ONNX")
    args = ap.parse_args(argv)
    from . import config as C
    conf = C.load_config(args.config)
    for key in ("data_root", "uid_list_file", "seed"):
        if getattr(args, key) is not None:
            conf[key] = getattr(args, key)
    kw = dict(conf["pipe_validation_kwargs"])                   # eta, guidance_scale, num_inference_steps
    if args.num_inference_steps is not None:
        kw["num_inference_steps"] = args.num_inference_steps
    if list(conf["views"]) != D.VIEWS or int(conf["validation_dataset"]["num_views"]) != 6:
        raise NotImplementedError("views: the six Wonder3D views in the shipped order only")
    if args.pretrained is None and os.path.isdir(str(conf["pretrained_model_name_or_path"])):
        args.pretrained = conf["pretrained_model_name_or_path"]
    args.data_root, args.uid_list_file, args.seed = conf["data_root"], conf["uid_list_file"], int(conf["seed"])
    rank, world, local = ddist.init()
    dev = torch.device("cuda", local)
    from ..mv import checkpoint as ck
    if args.pretrained:
        pipe = ck.load_pipeline(args.pretrained, dev)
    else:
        missing = [n for n, v in (("--unet_state_dict", args.unet_state_dict),
                                  ("--vae_state_dict", args.vae_state_dict),
                                  ("--image_encoder", args.image_encoder)) if not v]
        if missing and not args.random_init:
            raise SystemExit("mv: no weights for " + ", ".join(missing) + " (give --pretrained DIR or "
                             "the three files; --random_init runs on random weights and produces noise)")
        pipe = build_random_pipeline(dev, seed=0)
        if args.unet_state_dict:
            ck.load_unet(pipe.unet, args.unet_state_dict)
        if args.vae_state_dict:
            ck.load_vae(pipe.vae, args.vae_state_dict)
        if args.image_encoder:
            pipe.image_encoder = ck.load_image_encoder(args.image_encoder, dev)
    for m in (pipe.unet, pipe.vae, pipe.image_encoder):
        ddist.broadcast_module(m, 0)
    matting_fn = None
    if args.matting == "isnet":                             # mv.py:17-18: the session is made once
        if args.isnet_weights is None and not args.random_init:
            raise SystemExit("mv: --matting isnet needs --isnet_weights FILE (--random_init runs on "
                             "random weights and produces noise)")
        from ..mv import matting
        net = matting.load_isnet(args.isnet_weights, dev)
        ddist.broadcast_module(net, 0)
        matting_fn = matting.matting_fn(matting.IsnetSession(net, dev))
    uids = json.load(open(args.uid_list_file)) if args.all else [args.uid]
    for uid in ddist.shard(uids, rank, world):
        img_fn = os.path.join(args.data_root, uid, args.img_fn)
        if not os.path.exists(img_fn):
            img_fn = os.path.join(args.data_root, uid, "char/texture.png")
        single = Image.open(img_fn)                         # char/*.png are RGBA (mv.py:58)
        if single.mode != "RGBA":
            single = single.convert("RGBA")
        if uid in D.ADD_GRAY_UIDS:                          # mv.py:59-62
            single = D.add_gray(single)
        imgs, cam = D.mv_batch(single, args.pose_dir, size=int(conf["validation_dataset"]["img_wh"][0]))
        g = torch.Generator(device=dev).manual_seed(args.seed)
        # mv.py:70-86: the batch in the weight dtype (f16), output_type 'pt', one image per prompt
        out = pipe(imgs.to(dev, torch.float16), cam.to(dev, torch.float16), generator=g, output_type="pt",
                   num_images_per_prompt=1, **kw)
        D.write_mv_outputs(os.path.join(args.data_root, uid, args.save_folder), out[:6], out[6:], single,
                           res=tuple(conf["resolution"]), uid=uid, matting_fn=matting_fn)
        print(uid, flush=True)


if __name__ == "__main__":
    main()
