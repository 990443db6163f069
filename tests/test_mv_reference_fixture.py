"""tests/golden/mv_reference.npz = the REFERENCE's own UNetMV2DConditionModel (mvdiffusion/models/
*.py, unmodified, float64 on the CPU over oracle/stubs) at reduced width.  CPU checks:
  * oracle/mv_ref.UNetRef (the restatement every GPU test of M1-M6 is held to) reproduces the
    reference's output and per-block intermediates to 1e-9;
  * the product's module tree (drawingspinup_amd/mv/unet.py) has exactly the reference's
    state_dict key set and shapes for that configuration.
The HIP UNet is held to the same fixture in tests/test_gpu_unet.py (-m gpu)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import mv_ref as mr
from oracle import mv_weights

FIX = os.path.join(os.path.dirname(__file__), "golden", "mv_reference.npz")


@pytest.fixture(scope="module")
def fx():
    z = np.load(FIX)
    cfg = json.loads(str(z["cfg_json"]))
    names_shapes = [(str(n), tuple(int(v) for v in str(s).split(",")) if str(s) else ())
                    for n, s in zip(z["names"], z["shapes"])]
    return z, cfg, names_shapes


def fixture_inputs():
    sample = mv_weights.det_tensor("in.sample", (12, 8, 16, 16), 1.5).half().double()
    ctx = mv_weights.det_tensor("in.ctx", (12, 1, 768), 1.0).half().double()
    cam = mv_weights.det_tensor("in.cam", (12, 5), 3.0)
    cl = torch.cat([torch.sin(cam), torch.cos(cam)], -1).half().double()
    return sample, torch.tensor([487]), ctx, cl


def test_det_uniform_is_a_pure_integer_function():
    # KAT of the generator itself (splitmix64 of (index+1)*golden + crc32(key)*M1), so a fixture
    # regenerated on another machine uses the same parameters
    v = mv_weights.det_uniform("conv_in.weight", 4)
    assert v.dtype == np.float64 and np.all(np.abs(v) < 1)
    again = mv_weights.det_uniform("conv_in.weight", 4)
    assert np.array_equal(v, again)
    assert not np.array_equal(v, mv_weights.det_uniform("conv_in.bias", 4))
    u = mv_weights.det_uniform("x", 200000)
    assert abs(u.mean()) < 5e-3 and abs(u.var() - 1 / 3) < 5e-3


def test_oracle_unet_matches_the_reference_modules(fx):
    z, cfg, names_shapes = fx
    sd = mv_weights.synth_state_dict(names_shapes)
    ref = mr.UNetRef(sd, tuple(cfg["block_out_channels"]), tuple(cfg["down_block_types"]),
                     tuple(cfg["up_block_types"]), layers_per_block=cfg["layers_per_block"],
                     heads=cfg["attention_head_dim"], groups=cfg["norm_num_groups"],
                     num_views=cfg["num_views"], cd_attention_mid=cfg["cd_attention_mid"],
                     temb_dtype=torch.float32)
    ref.taps = {}
    out = ref(*fixture_inputs())
    want = torch.from_numpy(z["out"])
    assert float((out - want).abs().max()) < 1e-9
    keep = z["keep"]
    rename = {"tb.norm1": "down_blocks.0.attentions.0.transformer_blocks.0.norm1",
              "tb.attn1": "down_blocks.0.attentions.0.transformer_blocks.0.attn1",
              "tb.attn_joint_mid": "down_blocks.0.attentions.0.transformer_blocks.0.attn_joint_mid",
              "tb.attn2": "down_blocks.0.attentions.0.transformer_blocks.0.attn2",
              "tb.ff": "down_blocks.0.attentions.0.transformer_blocks.0.ff",
              "tb.out": "down_blocks.0.attentions.0.transformer_blocks.0.out"}
    checked = 0
    for key in z.files:
        if not key.startswith("tap."):
            continue
        name = rename.get(key[4:], key[4:])
        got = ref.taps[name][keep]
        w = torch.from_numpy(z[key]).double()
        # the fixture stores the intermediates in float32
        assert float((got - w).abs().max()) < 2e-6 * max(1.0, float(w.abs().max())), key
        checked += 1
    assert checked >= 20


def test_product_module_tree_has_the_reference_state_dict_layout(fx):
    _, cfg, names_shapes = fx
    from drawingspinup_amd.mv.unet import UNetMV2DConditionModel
    model = UNetMV2DConditionModel(
        sample_size=cfg["sample_size"], in_channels=cfg["in_channels"],
        out_channels=cfg["out_channels"], block_out_channels=tuple(cfg["block_out_channels"]),
        layers_per_block=cfg["layers_per_block"], cross_attention_dim=cfg["cross_attention_dim"],
        attention_head_dim=cfg["attention_head_dim"], norm_num_groups=cfg["norm_num_groups"],
        projection_class_embeddings_input_dim=cfg["projection_class_embeddings_input_dim"],
        num_views=cfg["num_views"], cd_attention_mid=cfg["cd_attention_mid"],
        down_block_types=tuple(cfg["down_block_types"]), up_block_types=tuple(cfg["up_block_types"]))
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    theirs = dict(names_shapes)
    assert set(mine) == set(theirs), (sorted(set(mine) ^ set(theirs))[:10])
    assert all(mine[k] == theirs[k] for k in theirs)
    # and it loads strictly
    model.load_state_dict({k: v.float() for k, v in mv_weights.synth_state_dict(names_shapes).items()},
                          strict=True)
