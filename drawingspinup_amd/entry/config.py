"""The reference's configuration surface for the entry points.

recon.py / mv.py load a YAML file through OmegaConf (recon.py:13-22, mv.py:21-26): `${a.b}`
interpolation, three custom resolvers (`calc_exp_lr_decay_rate`, `add`, `sub`, recon.py:13-15), an
(empty) dotlist of command-line overrides, `OmegaConf.resolve`.  OmegaConf is not a dependency
here: `load_config` does the same over PyYAML.  test_stage{1,2}.py / train_stage{1,2}.py read
`configs/config_stage{1,2}.yaml` with plain `yaml.load(...)['job']` (test_stage1.py:23-26).

The reference resolves `--config` relative to the working directory (`./configs/...yaml`, the
reference checkout).  When that file does not exist — the package used outside a reference
checkout — the values the reference ships are taken from BUILTIN below, keyed by the file's base
name; tests/test_entry_config.py holds BUILTIN to the reference's YAML files (resolved) key by key.
"""
import copy
import os
import re

import yaml

from ..nsr.model import Cfg

RESOLVERS = {                                                   # recon.py:13-15
    "calc_exp_lr_decay_rate": lambda factor, n: factor ** (1.0 / n),
    "add": lambda a, b: a + b,
    "sub": lambda a, b: a - b,
}
_INNER = re.compile(r"\$\{([^${}]*)\}")


def _lookup(root, path):
    node = root
    for part in path.split("."):
        node = node[int(part)] if isinstance(node, list) else node[part]
    return node


def _literal(text):
    text = text.strip()
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError:
        return text


def _eval(root, expr):
    """value of one `${...}` body whose own arguments are already plain text."""
    if ":" in expr:
        name, args = expr.split(":", 1)
        if name.strip() not in RESOLVERS:
            raise KeyError(f"unknown resolver `{name}` (the reference registers: {sorted(RESOLVERS)})")
        return RESOLVERS[name.strip()](*[_literal(a) for a in args.split(",")])
    return _resolve_value(root, _lookup(root, expr.strip()))


def _resolve_value(root, value):
    if not isinstance(value, str) or "${" not in value:
        return value
    while True:
        m = _INNER.search(value)
        if m is None:
            return value
        got = _eval(root, m.group(1))
        if m.start() == 0 and m.end() == len(value):
            return got                                          # keeps its type
        value = value[:m.start()] + str(got) + value[m.end():]


def _resolve_tree(root, node):
    if isinstance(node, dict):
        for k in list(node):
            node[k] = _resolve_tree(root, node[k])
        return node
    if isinstance(node, list):
        return [_resolve_tree(root, v) for v in node]
    return _resolve_value(root, node)


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _from_dotlist(cli_args):
    out = {}
    for item in cli_args or []:
        key, _, val = item.partition("=")
        node = out
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = _literal(val)
    return out


def _plain(node):
    """plain dict / list copy of a config tree (Cfg's attribute access confuses copy.deepcopy)."""
    if isinstance(node, dict):
        return {k: _plain(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        return [_plain(v) for v in node]
    return node


def builtin_name(path):
    return os.path.splitext(os.path.basename(path))[0]


def load_config(*yaml_files, cli_args=()):
    """recon.py:16-21 / mv.py:21-26: merge the files, then the dotlist, then resolve."""
    conf = {}
    for f in yaml_files:
        if os.path.isfile(f):
            with open(f) as fh:
                _merge(conf, yaml.safe_load(fh) or {})
        elif builtin_name(f) in BUILTIN:
            _merge(conf, copy.deepcopy(BUILTIN[builtin_name(f)]))
        else:
            raise FileNotFoundError(f"{f}: no such config file, and no built-in `{builtin_name(f)}`")
    _merge(conf, _from_dotlist(cli_args))
    return Cfg(_resolve_tree(conf, conf))


def load_stage_job(stage, path=None):
    """test_stage1.py:23-26: yaml.load(open('configs/config_stage1.yaml'))['job'].  The built-in
    copy stands in only for the implicit default file; an explicit `path` must exist."""
    if path is not None and not os.path.isfile(path):
        raise FileNotFoundError(f"{path}: no such stage-{stage} config file")
    path = path or f"configs/config_stage{stage}.yaml"
    if os.path.isfile(path):
        with open(path) as f:
            return yaml.load(f, Loader=yaml.FullLoader)["job"]
    return copy.deepcopy(BUILTIN[f"config_stage{stage}"]["job"])


# ------------------------------------------------------------------------------------------------
# adapters: YAML structure -> what OrthoNeuSSystem / NeuSModel consume
# ------------------------------------------------------------------------------------------------
def exp_lr_gamma(max_steps, constant_steps):
    """${calc_exp_lr_decay_rate:0.1,${sub:max_steps,constant_steps}} (recon.py:13).  A run that ends
    inside the constant phase (smoke runs: --max_steps <= constant_steps) never reaches the
    exponential scheduler: factor 1.0 instead of the resolver's division by zero / growing rate."""
    decay_steps = int(max_steps) - int(constant_steps)
    return 0.1 ** (1.0 / decay_steps) if decay_steps > 0 else 1.0


def nsr_configs(conf):
    """(model_config, system_config) of drawingspinup_amd.nsr.system.OrthoNeuSSystem from a resolved
    neuralangelo-ortho-wmask.yaml.  Options the HIP path does not implement raise instead of being
    ignored."""
    s = conf["system"]
    opt = s["optimizer"]
    if opt["name"] != "AdamW":
        raise NotImplementedError(f"system.optimizer.name = {opt['name']} (AdamW only)")
    sch = s["scheduler"]
    kinds = [x["name"] for x in sch["schedulers"]]
    if sch["name"] != "SequentialLR" or kinds != ["ConstantLR", "ExponentialLR"] or \
            list(sch["milestones"]) != [s["constant_steps"]] or sch["schedulers"][0]["args"]["factor"] != 1.0:
        raise NotImplementedError("system.scheduler: SequentialLR[ConstantLR(1.0), ExponentialLR] only")
    max_steps = int(conf["trainer"]["max_steps"])
    gamma = sch["schedulers"][1]["args"]["gamma"]
    want = exp_lr_gamma(max_steps, s["constant_steps"])
    if abs(gamma - want) > 1e-12:
        raise NotImplementedError(f"ExponentialLR gamma {gamma}: only calc_exp_lr_decay_rate(0.1, "
                                  "max_steps - constant_steps) is implemented")
    system_config = {
        "loss": dict(s["loss"]),
        "optimizer": {"lr": opt["args"]["lr"], "betas": tuple(opt["args"]["betas"]),
                      "eps": opt["args"]["eps"],
                      "params": {k: v["lr"] for k, v in opt["params"].items()}},
        "constant_steps": int(s["constant_steps"]), "max_steps": max_steps,
    }
    model_config = _plain(conf["model"])
    geo = model_config["geometry"]
    if geo["xyz_encoding_config"]["otype"] != "ProgressiveBandHashGrid" or \
            geo["mlp_network_config"]["otype"] != "VanillaMLP" or geo["grad_type"] != "finite_difference":
        raise NotImplementedError("model.geometry: ProgressiveBandHashGrid + VanillaMLP + "
                                  "finite_difference only")
    return model_config, system_config


def export_save_name(conf, global_step):
    """neus_ortho.py:183-194."""
    g = conf["model"]["geometry"]
    name = f"it{global_step}-{g['isosurface']['method']}{g['isosurface']['resolution']}-f{g['face_count']}"
    for flag, suffix in ((g["front_cutting"], "_c"), (g["remeshing"], "_r"),
                         (conf["export"]["thinning"], "_t"), (conf["export"]["smoothing"], "_s"),
                         (conf["export"]["color_back_projection"], "_cbp")):
        if flag:
            name += suffix
    return name


# ------------------------------------------------------------------------------------------------
# the values the reference ships (its YAML files, interpolations left in place)
# ------------------------------------------------------------------------------------------------
_DATA = "../dataset/AnimatedDrawings"
_GEN = {"use_bias": False, "tanh": True, "append_smoothers": True, "resnet_blocks": 7,
        "filters": [32, 64, 128, 128, 128, 64], "input_channels": 3}
_ADAM = {"type": "Adam", "args": {"lr": 0.0004, "betas": [0.9, 0.999], "weight_decay": 0.00001}}


def _stage(stage):
    return {"job": {
        "generator": {"type": "GeneratorJ_RIC" if stage == 1 else "GeneratorJ", "args": dict(_GEN)},
        "opt_generator": copy.deepcopy(_ADAM),
        "discriminator": {"type": "DiscriminatorN_IN", "args": {"num_filters": 12, "n_layers": 2}},
        "opt_discriminator": copy.deepcopy(_ADAM),
        "perception_loss": {"weight": 6.0, "perception_model": {
            "type": "PerceptualVGG19", "args": {"feature_layers": [0, 3, 5], "use_normalization": False}}},
        "trainer": {"batch_size": 40, "num_workers": 1, "epochs": 3 if stage == 1 else 2,
                    "reconstruction_weight": 4.0, "adversarial_weight": 0.5, "use_image_loss": True,
                    "reconstruction_criterion": "L1Loss", "adversarial_criterion": "MSELoss",
                    "log_interval": 1000, "patch_size": 32,
                    "pre_dir": "color" if stage == 1 else "res_stage1_mask_pos",
                    "post_name": "ffc_resnet_inpainted" if stage == 1 else "texture_with_bg"},
        "device": "cuda:0", "root_dir": _DATA + "/preprocessed"}}


BUILTIN = {
    "mvdiffusion-joint-ortho-6views": {
        "seed": 123456, "pretrained_model_name_or_path": "flamehaze1115/wonder3d-v1.0",
        "data_root": _DATA + "/preprocessed", "uid_list_file": _DATA + "/drawings_uids.json",
        "views": ["front", "front_right", "right", "back", "left", "front_left"],
        "resolution": [1024, 1024],
        "validation_dataset": {"num_views": 6, "bg_color": "white", "img_wh": [256, 256], "crop_size": -1},
        "pipe_validation_kwargs": {"eta": 1.0, "guidance_scale": 1.0, "num_inference_steps": 75},
    },
    "neuralangelo-ortho-wmask": {
        "seed": 123456,
        "dataset": {"name": "ortho", "cam_pose_dir": "./instant_nsr/datasets/fixed_poses",
                    "imSize": [1024, 1024], "data_root": _DATA + "/preprocessed",
                    "uid_list_file": _DATA + "/drawings_uids.json",
                    "thinning_uid_list_file": _DATA + "/drawings_uids_thinning.json",
                    "load_front_mask": True},
        "export": {"chunk_size": 2097152, "ortho_scale": 1.35, "thinning": True, "smoothing": True,
                   "thinning_type": "double", "shearing": True, "color_back_projection": True,
                   "export_uv": False},
        "model": {
            "name": "neus", "radius": 1.0, "num_samples_per_ray": 1024, "train_num_rays": 256,
            "max_train_num_rays": 8192, "grid_prune": True, "grid_prune_occ_thre": 0.001,
            "dynamic_ray_sampling": True, "batch_image_sampling": True, "randomized": True,
            "ray_chunk": 2048, "cos_anneal_end": 20000,
            "variance": {"init_val": 0.3, "modulate": False},
            "geometry": {
                "name": "volume-sdf", "radius": "${model.radius}", "feature_dim": 13,
                "grad_type": "finite_difference", "finite_difference_eps": "progressive",
                "front_cutting": True, "remeshing": True, "face_count": 50000,
                "isosurface": {"method": "mc", "resolution": 512, "chunk": 2097152, "threshold": 0.0},
                "xyz_encoding_config": {
                    "otype": "ProgressiveBandHashGrid", "n_levels": 10, "n_features_per_level": 2,
                    "log2_hashmap_size": 19, "base_resolution": 32,
                    "per_level_scale": 1.3195079107728942, "include_xyz": True, "start_level": 4,
                    "start_step": 0, "update_steps": 1000},
                "mlp_network_config": {
                    "otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none",
                    "n_neurons": 64, "n_hidden_layers": 1, "sphere_init": True,
                    "sphere_init_radius": 0.5, "weight_norm": True}},
            "texture": {
                "name": "volume-radiance",
                "input_feature_dim": "${add:${model.geometry.feature_dim},3}",
                "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 4},
                "mlp_network_config": {"otype": "VanillaMLP", "activation": "ReLU",
                                       "output_activation": "none", "n_neurons": 64,
                                       "n_hidden_layers": 2},
                "color_activation": "sigmoid"}},
        "system": {
            "name": "ortho-neus-system",
            "loss": {"lambda_rgb_mse": 0.5, "lambda_rgb_l1": 0.0, "lambda_mask": 1.0,
                     "lambda_eikonal": 0.2, "lambda_normal": 1.0, "lambda_3d_normal_smooth": 1.0,
                     "lambda_sparsity": 0.5, "sparsity_scale": 100.0, "geo_aware": True,
                     "rgb_p_ratio": 0.8, "normal_p_ratio": 0.8, "mask_p_ratio": 0.9},
            "optimizer": {"name": "AdamW", "args": {"lr": 0.01, "betas": [0.9, 0.99], "eps": 1e-15},
                          "params": {"geometry": {"lr": 0.001}, "texture": {"lr": 0.01},
                                     "variance": {"lr": 0.001}}},
            "constant_steps": 500,
            "scheduler": {
                "name": "SequentialLR", "interval": "step", "milestones": ["${system.constant_steps}"],
                "schedulers": [
                    {"name": "ConstantLR", "args": {"factor": 1.0, "total_iters": "${system.constant_steps}"}},
                    {"name": "ExponentialLR", "args": {
                        "gamma": "${calc_exp_lr_decay_rate:0.1,${sub:${trainer.max_steps},${system.constant_steps}}}"}}]}},
        "checkpoint": {"save_top_k": -1, "every_n_train_steps": "${trainer.max_steps}"},
        "trainer": {"max_steps": 3000, "log_every_n_steps": 100, "num_sanity_val_steps": 0,
                    "val_check_interval": 10000, "limit_train_batches": 1.0, "limit_val_batches": 2,
                    "enable_progress_bar": True, "precision": 16},
    },
    "config_stage1": _stage(1),
    "config_stage2": _stage(2),
}
