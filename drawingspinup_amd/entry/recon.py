"""`python recon.py --uid U [--all]` of the reference (2_charactor_reconstructor/recon.py:44-62):
3000 optimisation steps, then the export (neus_ortho.py:183-200): 2 x 512^3 SDF volumes, constrained
smoothing, front-mask cutting (char/mask.png rotated as ortho.py:155-156), marching cubes, vertex
colours, written as <uid>/mesh/it3000-mc512-f50000_c[_r][_t][_s][_cbp].obj.  The reference's config
switches are flags here: quadric decimation of the fine mesh (`--remeshing`, `_r`: host function of
the library), the thinning deformation (`--thinning`, `_t`: nsr/thinning.py, only for the uids of
`--thinning_uid_list_file` as recon.py:53-65 does), Laplacian smoothing (`--smoothing`, `_s`),
colour back-projection (`--color_back_projection`, `_cbp`: device kernels, nsr/mesh_post.py) and
shear (`--shearing`)."""
import argparse
import json
import os

import numpy as np
import torch

from .. import dist as ddist
from ..nsr.system import OrthoNeuSSystem
from . import data as D


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--uid", default="0dd66be9d0534b93a092d8c4c4dfd30a")
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--data_root", default="../dataset/AnimatedDrawings/preprocessed")
    ap.add_argument("--uid_list_file", default="../dataset/AnimatedDrawings/drawings_uids.json")
    ap.add_argument("--pose_dir", default=None, help=".../instant_nsr/datasets/fixed_poses")
    ap.add_argument("--max_steps", type=int, default=3000)
    ap.add_argument("--seed", type=int, default=123456)          # recon.py:30
    # model.geometry.remeshing / face_count (configs/neuralangelo-ortho-wmask.yaml:45-46)
    ap.add_argument("--remeshing", action="store_true")
    ap.add_argument("--face_count", type=int, default=50000)
    # export.thinning / thinning_type (yaml:15-17); applied to the uids of the thinning list only
    ap.add_argument("--thinning", action="store_true")
    ap.add_argument("--thinning_type", default="double", choices=["double", "front", "back"])
    ap.add_argument("--thinning_uid_list_file", default=None)
    # export.smoothing / export.shearing of the reference's config (mesh_utils.py:42-58); all of
    # these are off by default here (the reference's yaml has them on)
    ap.add_argument("--smoothing", action="store_true")
    ap.add_argument("--shearing", action="store_true")
    # export.color_back_projection (coloring_utils.py:91-138) from <uid>/mv/{color,mask}/*.png
    ap.add_argument("--color_back_projection", action="store_true")
    args = ap.parse_args(argv)
    rank, world, local = ddist.init()
    dev = torch.device("cuda", local)
    uids = json.load(open(args.uid_list_file)) if args.all else [args.uid]
    thinning_uids = None
    if args.thinning and args.thinning_uid_list_file:
        thinning_uids = set(json.load(open(args.thinning_uid_list_file)))
    for uid in ddist.shard(uids, rank, world):
        ds = D.load_mv_prediction(os.path.join(args.data_root, uid, "mv"), dev, args.pose_dir, uid=uid)
        system = OrthoNeuSSystem(device=dev, seed=args.seed)
        system.fit(ds, max_steps=args.max_steps, log_every=500)
        front = fm = None
        fm_path = os.path.join(args.data_root, uid, "char", "mask.png")
        if os.path.isfile(fm_path):
            from PIL import Image
            fm = np.array(Image.open(fm_path).convert("L"))
            front = torch.from_numpy(np.ascontiguousarray(np.rot90(fm, k=-1))).to(dev)   # cv2.ROTATE_90_CLOCKWISE
        mesh = system.export_mesh(front, face_count=args.face_count if args.remeshing else None)
        thin = args.thinning and fm is not None and (thinning_uids is None or uid in thinning_uids)
        out = os.path.join(args.data_root, uid, "mesh")
        os.makedirs(out, exist_ok=True)
        from ..nsr.mesh import save_obj
        name = system.export_name(front is not None) + ("_r" if args.remeshing else "") \
            + ("_t" if thin else "") + ("_s" if args.smoothing else "") \
            + ("_cbp" if args.color_back_projection else "")                     # neus_ortho.py:183-194
        cbp = None
        if args.color_back_projection:
            from PIL import Image
            mv = os.path.join(args.data_root, uid, "mv")
            big = lambda sub, view, mode: torch.from_numpy(np.array(
                Image.open(os.path.join(mv, sub, view + ".png")).convert(mode)
                .resize((2048, 2048), Image.LANCZOS))).to(dev)                   # coloring_utils.py:62,100
            cbp = {"color_front": big("color", "front", "RGB"), "color_back": big("color", "back", "RGB"),
                   "mask_front": big("mask", "front", "L")}
        save_obj(os.path.join(out, name + ".obj"), mesh["verts"], mesh["faces"], mesh["vert_colors"],
                 smoothing=args.smoothing, shearing=args.shearing, color_back_projection=cbp,
                 thinning={"mask": fm, "type": args.thinning_type} if thin else None)
        torch.save(system.model.state_dict(), os.path.join(out, f"it{system.global_step}.ckpt"))
        print(uid, flush=True)


if __name__ == "__main__":
    main()
