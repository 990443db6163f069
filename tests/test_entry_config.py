"""The configuration surface of the entry points (drawingspinup_amd/entry/config.py) against the
reference's YAML files (tests/golden/config_reference.json, make_config_golden.py) and the
semantics of OmegaConf's interpolation as recon.py:13-21 uses it."""
import json
import os

import pytest

from drawingspinup_amd.entry import config as C

FIX = os.path.join(os.path.dirname(__file__), "golden", "config_reference.json")


@pytest.fixture(scope="module")
def ref():
    return json.load(open(FIX))


@pytest.mark.parametrize("name", ["neuralangelo-ortho-wmask", "mvdiffusion-joint-ortho-6views"])
def test_builtin_equals_the_reference_yaml(ref, name):
    got = C._plain(C.load_config(f"./configs/{name}.yaml"))           # no such file here -> BUILTIN
    assert got == ref[name]


@pytest.mark.parametrize("stage", [1, 2])
def test_builtin_stage_jobs_equal_the_reference_yaml(ref, stage):
    assert C.load_stage_job(stage) == ref[f"config_stage{stage}"]["job"]


def test_interpolation_and_resolvers_hand_checked(tmp_path):
    conf = C.load_config("neuralangelo-ortho-wmask.yaml")
    # ${model.radius}; ${add:${model.geometry.feature_dim},3}; nested calc_exp_lr_decay_rate/sub
    assert conf.model.geometry.radius == 1.0 and isinstance(conf.model.geometry.radius, float)
    assert conf.model.texture.input_feature_dim == 16
    assert conf.system.scheduler.milestones == [500]
    gamma = conf.system.scheduler.schedulers[1]["args"]["gamma"]
    assert gamma == 0.1 ** (1.0 / 2500) and abs(gamma ** 2500 - 0.1) < 1e-12
    assert conf.checkpoint.every_n_train_steps == 3000
    assert conf.system.optimizer.args.eps == 1e-15                    # YAML `1.e-15` is a float
    # a file on disk + a dotlist override (OmegaConf.from_cli), resolved after the merge
    p = tmp_path / "c.yaml"
    p.write_text("a: {b: 2, c: '${a.b}'}\nd: '${add:${a.b},${a.c}}'\ne: 'x${a.b}y'\n")
    c = C.load_config(str(p), cli_args=["a.b=5"])
    assert c.a.c == 5 and c.d == 10 and c.e == "x5y"
    with pytest.raises(KeyError):
        C.load_config(str(p), cli_args=["d=${mul:1,2}"])
    with pytest.raises(FileNotFoundError):
        C.load_config("no-such-config.yaml")


def test_nsr_adapter_and_save_name():
    from drawingspinup_amd.nsr.model import DEFAULT_MODEL_CONFIG
    from drawingspinup_amd.nsr.system import DEFAULT_SYSTEM_CONFIG
    conf = C.load_config("neuralangelo-ortho-wmask.yaml")
    model_config, system_config = C.nsr_configs(conf)
    assert system_config == C._plain(DEFAULT_SYSTEM_CONFIG) or \
        {**system_config, "optimizer": {**system_config["optimizer"],
                                        "betas": list(system_config["optimizer"]["betas"])}} == \
        C._plain(DEFAULT_SYSTEM_CONFIG)
    for k, v in C._plain(DEFAULT_MODEL_CONFIG).items():                # the yaml holds a superset
        if isinstance(v, dict):
            for kk, vv in v.items():
                assert model_config[k][kk] == vv, (k, kk)
        else:
            assert model_config[k] == v, k
    # neus_ortho.py:183-194 with the shipped switches
    assert C.export_save_name(conf, 3000) == "it3000-mc512-f50000_c_r_t_s_cbp"
    conf["export"]["thinning"] = False                                 # recon.py:58-59,63-64
    assert C.export_save_name(conf, 3000) == "it3000-mc512-f50000_c_r_s_cbp"
    conf["system"]["optimizer"]["name"] = "SGD"
    with pytest.raises(NotImplementedError):
        C.nsr_configs(conf)


def test_recon_cli_defaults_are_the_yaml(tmp_path):
    from drawingspinup_amd.entry import recon
    lst = tmp_path / "thin.json"
    lst.write_text(json.dumps(["u2"]))
    args, conf = recon.parse(["--uid", "u1", "--thinning_uid_list_file", str(lst)])
    ex, geo = conf["export"], conf["model"]["geometry"]
    assert args.config == "./configs/neuralangelo-ortho-wmask.yaml"
    assert geo["remeshing"] and geo["face_count"] == 50000 and geo["front_cutting"]
    assert ex["thinning"] and ex["smoothing"] and ex["shearing"] and ex["color_back_projection"]
    assert recon.uids_and_thinning(args, conf) == [("u1", False)]      # not in the thinning list
    args, conf = recon.parse(["--uid", "u2", "--thinning_uid_list_file", str(lst)])
    assert recon.uids_and_thinning(args, conf) == [("u2", True)]
    # opt-outs and overrides
    args, conf = recon.parse(["--uid", "u2", "--no-remeshing", "--no-thinning", "--max_steps", "100",
                              "model.geometry.isosurface.resolution=128"])
    assert not conf["model"]["geometry"]["remeshing"] and not conf["export"]["thinning"]
    assert conf["trainer"]["max_steps"] == 100 and conf["model"]["geometry"]["isosurface"]["resolution"] == 128
    assert recon.uids_and_thinning(args, conf) == [("u2", False)]
    assert C.export_save_name(conf, 100) == "it100-mc128-f50000_c_s_cbp"
    # the reference opens the thinning list unconditionally when export.thinning is on
    args, conf = recon.parse(["--uid", "u1", "--thinning_uid_list_file", str(tmp_path / "missing.json")])
    with pytest.raises(FileNotFoundError):
        recon.uids_and_thinning(args, conf)


def test_explicit_stage_config_must_exist(tmp_path):
    """A mistyped --config must not silently train with the shipped defaults."""
    import pytest
    with pytest.raises(FileNotFoundError):
        C.load_stage_job(1, str(tmp_path / "config_stage1_typo.yaml"))


def test_recon_max_steps_inside_the_constant_phase_keeps_the_rate():
    """--max_steps <= system.constant_steps (smoke runs): no division by zero, no growing rate."""
    from drawingspinup_amd.entry import recon
    for m in ("500", "100"):
        _, conf = recon.parse(["--uid", "u", "--max_steps", m])
        assert conf["system"]["scheduler"]["schedulers"][1]["args"]["gamma"] == 1.0
        C.nsr_configs(conf)                                   # accepted by the adapter
    _, conf = recon.parse(["--uid", "u", "--max_steps", "600"])
    assert abs(conf["system"]["scheduler"]["schedulers"][1]["args"]["gamma"] - 0.1 ** (1 / 100)) < 1e-12
