"""`python recon.py --uid U [--all]` of the reference (2_charactor_reconstructor/recon.py:44-62):
3000 optimisation steps, then the export (neus_ortho.py:183-200): 2 x 512^3 SDF volumes, constrained
smoothing, front-mask cutting (char/mask.png rotated as ortho.py:155-156), marching cubes, vertex
colours, written as <uid>/mesh/it3000-mc512-f50000_c[_s][_cbp].obj.  Of save_mesh's steps, Laplacian
smoothing (`--smoothing`, `_s`), colour back-projection (`--color_back_projection`, `_cbp`: device
kernels, nsr/mesh_post.py) and shear (`--shearing`) are available; quadric decimation (`_r`) and
the thinning deformation (`_t`) are not."""
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
    # export.smoothing / export.shearing of the reference's config (mesh_utils.py:42-58); off by
    # default here: they are host-side steps on the full-resolution mesh (the reference decimates to
    # 50 000 faces first, which needs trimesh); smoothing adds the reference's `_s` to the save name
    ap.add_argument("--smoothing", action="store_true")
    ap.add_argument("--shearing", action="store_true")
    # export.color_back_projection (coloring_utils.py:91-138) from <uid>/mv/{color,mask}/*.png
    ap.add_argument("--color_back_projection", action="store_true")
    args = ap.parse_args(argv)
    rank, world, local = ddist.init()
    dev = torch.device("cuda", local)
    uids = json.load(open(args.uid_list_file)) if args.all else [args.uid]
    for uid in ddist.shard(uids, rank, world):
        ds = D.load_mv_prediction(os.path.join(args.data_root, uid, "mv"), dev, args.pose_dir, uid=uid)
        system = OrthoNeuSSystem(device=dev, seed=args.seed)
        system.fit(ds, max_steps=args.max_steps, log_every=500)
        front = None
        fm_path = os.path.join(args.data_root, uid, "char", "mask.png")
        if os.path.isfile(fm_path):
            from PIL import Image
            fm = np.array(Image.open(fm_path).convert("L"))
            front = torch.from_numpy(np.ascontiguousarray(np.rot90(fm, k=-1))).to(dev)   # cv2.ROTATE_90_CLOCKWISE
        mesh = system.export_mesh(front)
        out = os.path.join(args.data_root, uid, "mesh")
        os.makedirs(out, exist_ok=True)
        from ..nsr.mesh import save_obj
        name = system.export_name(front is not None) + ("_s" if args.smoothing else "") \
            + ("_cbp" if args.color_back_projection else "")                     # neus_ortho.py:190-195
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
                 smoothing=args.smoothing, shearing=args.shearing, color_back_projection=cbp)
        torch.save(system.model.state_dict(), os.path.join(out, f"it{system.global_step}.ckpt"))
        print(uid, flush=True)


if __name__ == "__main__":
    main()
