"""`python recon.py [--config configs/neuralangelo-ortho-wmask.yaml] --uid U [--all]` of the
reference (2_charactor_reconstructor/recon.py:24-65): the YAML file is the source of every setting
(loaded as recon.py:16-21 does, entry/config.py), so a bare `recon --uid U` runs what the
reference runs: 3000 optimisation steps, then the export of neus_ortho.py:183-200 with the
shipped switches ON — front-mask cutting (`_c`), quadric remeshing to `face_count` faces (`_r`),
thinning for the uids of `dataset.thinning_uid_list_file` only (`_t`, recon.py:53-65), Laplacian
smoothing (`_s`), shearing, colour back-projection (`_cbp`) — written as
<data_root>/<uid>/mesh/it3000-mc512-f50000_c_r[_t]_s_cbp.obj.

The extra flags below are overrides for runs outside a reference checkout (tests, synthetic
data); left unset, the config decides.
"""
import argparse
import json
import os

import numpy as np
import torch

from .. import dist as ddist
from ..nsr.system import OrthoNeuSSystem
from . import config as C
from . import data as D


def _bool_flag(ap, name, help=None):
    g = ap.add_mutually_exclusive_group()
    g.add_argument("--" + name, dest=name, action="store_true", default=None, help=help)
    g.add_argument("--no-" + name, dest=name, action="store_false")


def parse(argv=None):
    ap = argparse.ArgumentParser(description="reconstruction")
    ap.add_argument("--config", default="./configs/neuralangelo-ortho-wmask.yaml")      # recon.py:47
    ap.add_argument("--uid", default="0dd66be9d0534b93a092d8c4c4dfd30a", help="image uid")
    ap.add_argument("--all", action="store_true", help="process all examples")
    # overrides (not in the reference's CLI; None = take the config's value)
    ap.add_argument("--data_root", default=None)
    ap.add_argument("--uid_list_file", default=None)
    ap.add_argument("--thinning_uid_list_file", default=None)
    ap.add_argument("--pose_dir", default=None, help="dataset.cam_pose_dir")
    ap.add_argument("--max_steps", type=int, default=None, help="trainer.max_steps")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--face_count", type=int, default=None)
    ap.add_argument("--resolution", type=int, default=None, help="model.geometry.isosurface.resolution")
    ap.add_argument("--thinning_type", default=None, choices=["double", "front", "back"])
    for name in ("remeshing", "thinning", "smoothing", "shearing", "color_back_projection"):
        _bool_flag(ap, name)
    ap.add_argument("overrides", nargs="*", help="OmegaConf-style dotlist, e.g. trainer.max_steps=100")
    args = ap.parse_args(argv)
    conf = C.load_config(args.config, cli_args=args.overrides)
    ds, ex, geo = conf["dataset"], conf["export"], conf["model"]["geometry"]
    for key, dst, name in (("data_root", ds, "data_root"), ("uid_list_file", ds, "uid_list_file"),
                           ("thinning_uid_list_file", ds, "thinning_uid_list_file"),
                           ("pose_dir", ds, "cam_pose_dir"), ("seed", conf, "seed"),
                           ("face_count", geo, "face_count"), ("remeshing", geo, "remeshing"),
                           ("thinning", ex, "thinning"), ("smoothing", ex, "smoothing"),
                           ("shearing", ex, "shearing"), ("thinning_type", ex, "thinning_type"),
                           ("color_back_projection", ex, "color_back_projection")):
        if getattr(args, key) is not None:
            dst[name] = getattr(args, key)
    if args.max_steps is not None:
        conf["trainer"]["max_steps"] = args.max_steps
        sch = conf["system"]["scheduler"]["schedulers"][1]["args"]        # ${calc_exp_lr_decay_rate:...}
        sch["gamma"] = C.exp_lr_gamma(args.max_steps, conf["system"]["constant_steps"])
    if args.resolution is not None:
        geo["isosurface"]["resolution"] = args.resolution
    return args, conf


def uids_and_thinning(args, conf):
    """recon.py:52-65: the uid list and, per uid, whether export.thinning applies.
    Deliberate difference under --all: the reference sets `config.export.thinning = False` at the
    first uid outside the thinning list and never restores it, so every LATER uid is exported
    without thinning (and without the `_t` suffix) whatever the list says — an order-dependent
    side effect that cannot survive sharding the uid list over ranks.  Here the list decides per
    uid.  A single --uid run (the documented use, README step 2) is identical."""
    ds = conf["dataset"]
    if conf["export"]["thinning"]:
        with open(ds["thinning_uid_list_file"]) as f:             # the reference opens it unconditionally
            thinning_uids = set(json.load(f))
    else:
        thinning_uids = set()
    if args.all:
        with open(ds["uid_list_file"]) as f:
            uids = json.load(f)
    else:
        uids = [args.uid]
    return [(u, bool(conf["export"]["thinning"]) and u in thinning_uids) for u in uids]


def recon(uid, thinning, conf, dev):
    ds, ex, geo = conf["dataset"], conf["export"], conf["model"]["geometry"]
    mv = os.path.join(ds["data_root"], uid, "mv")                    # config.dataset.input_dir
    out = os.path.join(ds["data_root"], uid, "mesh")                 # config.export.output_dir
    pose_dir = ds["cam_pose_dir"] if os.path.isdir(str(ds["cam_pose_dir"])) else None
    data = D.load_mv_prediction(mv, dev, pose_dir, size=tuple(ds["imSize"]), uid=uid)
    model_config, system_config = C.nsr_configs(conf)
    system = OrthoNeuSSystem(model_config, system_config, device=dev, seed=int(conf["seed"]))
    system.fit(data, max_steps=int(conf["trainer"]["max_steps"]),
               log_every=int(conf["trainer"].get("log_every_n_steps", 0) or 0) * 5)
    front = fm = None
    if geo["front_cutting"]:                                         # dataset.load_front_mask (recon.py:29)
        from PIL import Image
        fm = np.array(Image.open(os.path.join(ds["data_root"], uid, "char", "mask.png")).convert("L"))
        front = torch.from_numpy(np.ascontiguousarray(np.rot90(fm, k=-1))).to(dev)   # cv2.ROTATE_90_CLOCKWISE
    cbp_on = bool(ex["color_back_projection"])
    mesh = system.export_mesh(front, face_count=int(geo["face_count"]) if geo["remeshing"] else None,
                              with_colors=not cbp_on)                # neus.py:224-236: rgb = None with cbp
    if thinning and fm is None:
        raise RuntimeError("export.thinning needs char/mask.png (thinning_utils.py:196-200)")
    run = C.Cfg(C._plain(conf))
    run["export"]["thinning"] = thinning
    name = C.export_save_name(run, system.global_step)
    cbp = None
    if cbp_on:
        from PIL import Image

        def big(sub, view, mode):                                    # coloring_utils.py:62,100
            im = Image.open(os.path.join(mv, sub, view + ".png")).convert(mode)
            return torch.from_numpy(np.array(im.resize((2048, 2048), Image.LANCZOS))).to(dev)
        cbp = {"color_front": big("color", "front", "RGB"), "color_back": big("color", "back", "RGB"),
               "mask_front": big("mask", "front", "L")}
    from ..nsr.mesh import save_obj
    os.makedirs(out, exist_ok=True)
    path = save_obj(os.path.join(out, name + ".obj"), mesh["verts"], mesh["faces"], mesh["vert_colors"],
                    ortho_scale=float(ex["ortho_scale"]), smoothing=bool(ex["smoothing"]),
                    shearing=bool(ex["shearing"]), color_back_projection=cbp,
                    thinning={"mask": fm, "type": ex["thinning_type"]} if thinning else None)
    torch.save(system.model.state_dict(), os.path.join(out, f"it{system.global_step}.ckpt"))
    return path


def main(argv=None):
    args, conf = parse(argv)
    rank, world, local = ddist.init()
    dev = torch.device("cuda", local)
    work = uids_and_thinning(args, conf)
    # one reconstruction after the other: the step driver's side stream at normal priority (include/dsu_hip.h —
    # at a non-default priority every fourth and later reconstruction of a process runs 0.8 s slower)
    from .. import _lib
    _lib.check(_lib.lib().dsu_set_nsr_side_stream_priority(2), "dsu_set_nsr_side_stream_priority")
    for uid, thinning in ddist.shard(work, rank, world):
        recon(uid, thinning, conf, dev)
        print(uid, flush=True)


if __name__ == "__main__":
    main()
