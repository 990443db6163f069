"""tests/golden/config_reference.json = the reference's four YAML files, loaded and resolved:
configs/neuralangelo-ortho-wmask.yaml and configs/mvdiffusion-joint-ortho-6views.yaml the way
recon.py:16-21 / mv.py:21-26 load them (interpolations + the three resolvers; OmegaConf is not
installable here, drawingspinup_amd/entry/config.py does its job over PyYAML, and the resolved
values are checked by hand in tests/test_entry_config.py), configs/config_stage{1,2}.yaml the way
test_stage1.py:23-26 does (plain yaml.load(...)['job']).

    python tests/golden/make_config_golden.py          # needs /root/reference
"""
import json
import os
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from drawingspinup_amd.entry import config as C  # noqa: E402

R2 = "/root/reference/2_charactor_reconstructor/configs/"
R3 = "/root/reference/3_style_translator/configs/"
ref = {
    "neuralangelo-ortho-wmask": C._plain(C.load_config(R2 + "neuralangelo-ortho-wmask.yaml")),
    "mvdiffusion-joint-ortho-6views": C._plain(C.load_config(R2 + "mvdiffusion-joint-ortho-6views.yaml")),
    "config_stage1": {"job": yaml.load(open(R3 + "config_stage1.yaml"), Loader=yaml.FullLoader)["job"]},
    "config_stage2": {"job": yaml.load(open(R3 + "config_stage2.yaml"), Loader=yaml.FullLoader)["job"]},
}
with open(os.path.join(ROOT, "tests", "golden", "config_reference.json"), "w") as f:
    json.dump(ref, f, indent=1, sort_keys=True)
print("wrote config_reference.json")
