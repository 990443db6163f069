"""Generate tests/golden/mv_reference.npz by running the REFERENCE's own mvdiffusion modules
(2_charactor_reconstructor/mvdiffusion/models/{unet_mv2d_condition,unet_mv2d_blocks,
transformer_mv2d}.py) UNMODIFIED, on the CPU in float64, in this container.

    python tests/golden/make_mv_reference_golden.py          # needs /root/reference

diffusers==0.19.3 and xformers==0.0.17 are absent and not installable; `oracle/stubs/` stands in
for exactly the symbols those three files import (see oracle/stubs/README.md).  The fixture thus
pins everything that is the reference's OWN code on rows M1-M6 of SURVEY.md §8(a):
  UNetMV2DConditionModel.__init__/forward (:228-374, :760-1054): time + class-embedding add,
    skip bookkeeping, block order, conv_norm_out/conv_out;
  CrossAttnDownBlockMV2D / CrossAttnUpBlockMV2D / UNetMidBlockMV2DCrossAttn (unet_mv2d_blocks.py);
  TransformerMV2DModel.forward (GroupNorm eps 1e-6, 1x1 proj_in/out, NCHW<->tokens, residual);
  BasicMVTransformerBlock.forward (attn1 -> joint_mid -> attn2 -> FF, norm placement);
  XFormersMVAttnProcessor / XFormersJointAttnProcessor (K/V regrouping over views / domains),
    cross-checked here against the reference's non-xformers MVAttnProcessor/JointAttnProcessor;
  the state_dict key set and shapes of the Wonder3D joint configuration.
The leaf arithmetic of the diffusers building blocks (ResnetBlock2D, Attention projections, GEGLU,
Timesteps, Down/Upsample2D) is served by the stubs: "reference graph pinned, op unpinned".

Width is reduced (80/160/320/320 channels, 2 heads of 40/80/160/160, 8 norm groups, 16x16
latents) so that the float64 run takes seconds; the architecture is the full one (4 levels,
2 layers per block, joint attention in the middle of every block, 6 views x 2 domains).
Parameters are regenerated from their names (oracle/mv_weights.py), not stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/2_charactor_reconstructor"
sys.path.insert(0, os.path.join(ROOT, "oracle", "stubs"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from oracle import mv_weights  # noqa: E402
from mvdiffusion.models import transformer_mv2d as tmv  # noqa: E402
from mvdiffusion.models.unet_mv2d_condition import UNetMV2DConditionModel  # noqa: E402

CFG = dict(sample_size=16, in_channels=8, out_channels=4, block_out_channels=(80, 160, 320, 320),
           layers_per_block=2, attention_head_dim=2, norm_num_groups=8, cross_attention_dim=768,
           class_embed_type="projection", projection_class_embeddings_input_dim=10, num_views=6,
           cd_attention_mid=True, cd_attention_last=False, multiview_attention=True,
           sparse_mv_attention=False, mvcd_attention=False,
           down_block_types=("CrossAttnDownBlockMV2D", "CrossAttnDownBlockMV2D",
                             "CrossAttnDownBlockMV2D", "DownBlock2D"),
           up_block_types=("UpBlock2D", "CrossAttnUpBlockMV2D", "CrossAttnUpBlockMV2D",
                           "CrossAttnUpBlockMV2D"))
KEEP = [0, 5, 6, 11]          # batch rows of the intermediates that are stored (2 per domain)


def inputs():
    sample = mv_weights.det_tensor("in.sample", (12, 8, 16, 16), 1.5).half().double()
    ctx = mv_weights.det_tensor("in.ctx", (12, 1, 768), 1.0).half().double()
    # class labels as mv.py:73-75 + pipeline:271-283 build them: sin|cos of (e, de, da, 2 task bits)
    cam = mv_weights.det_tensor("in.cam", (12, 5), 3.0)
    cl = torch.cat([torch.sin(cam), torch.cos(cam)], -1).half().double()
    return sample, torch.tensor([487]), ctx, cl


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    unet = UNetMV2DConditionModel(**CFG).double().eval()
    names_shapes = [(k, tuple(v.shape)) for k, v in unet.state_dict().items()]
    sd = mv_weights.synth_state_dict(names_shapes)
    unet.load_state_dict(sd, strict=True)

    taps = {}

    def tap(name, pick=lambda o: o):
        def hook(_m, _i, out):
            taps[name] = pick(out).detach().clone()
        return hook

    first = lambda o: o[0]                                        # noqa: E731
    unet.time_embedding.register_forward_hook(tap("time_embedding"))
    unet.class_embedding.register_forward_hook(tap("class_embedding"))
    unet.conv_in.register_forward_hook(tap("conv_in"))
    for i, b in enumerate(unet.down_blocks):
        b.register_forward_hook(tap(f"down_blocks.{i}", first))
    unet.mid_block.register_forward_hook(tap("mid_block"))
    for i, b in enumerate(unet.up_blocks):
        b.register_forward_hook(tap(f"up_blocks.{i}"))
    unet.down_blocks[0].resnets[0].register_forward_hook(tap("down_blocks.0.resnets.0"))
    tb = unet.down_blocks[0].attentions[0].transformer_blocks[0]
    tb.norm1.register_forward_hook(tap("tb.norm1"))
    tb.attn1.register_forward_hook(tap("tb.attn1"))
    tb.attn_joint_mid.register_forward_hook(tap("tb.attn_joint_mid"))
    tb.attn2.register_forward_hook(tap("tb.attn2"))
    tb.ff.register_forward_hook(tap("tb.ff"))
    tb.register_forward_hook(tap("tb.out"))
    unet.down_blocks[0].attentions[0].register_forward_hook(tap("down_blocks.0.attentions.0", first))

    sample, t, ctx, cl = inputs()
    # the reference's default processors (MVAttnProcessor / JointAttnProcessor) are what the
    # constructor installs; BasicMVTransformerBlock.forward passes `sparse_mv_attention=` which
    # MVAttnProcessor.__call__ does not accept (transformer_mv2d.py:556-565 vs :645-654), so the
    # reference only runs with the xformers processors mv.py:186-188 selects:
    assert isinstance(tb.attn1.processor, tmv.MVAttnProcessor)
    assert isinstance(tb.attn_joint_mid.processor, tmv.JointAttnProcessor)
    unet.enable_xformers_memory_efficient_attention()
    assert isinstance(tb.attn1.processor, tmv.XFormersMVAttnProcessor)
    assert isinstance(tb.attn_joint_mid.processor, tmv.XFormersJointAttnProcessor)
    out = unet(sample, t, encoder_hidden_states=ctx, class_labels=cl).sample
    # cross-check of the two processor families of the reference on one block's real input
    # (same K/V regrouping written twice in the reference: repeat_interleave vs my_repeat)
    x = mv_weights.det_tensor("in.tokens", (12, 256, 80), 1.0)
    a = tmv.MVAttnProcessor()(tb.attn1, x, num_views=6, multiview_attention=True)
    b = tmv.XFormersMVAttnProcessor()(tb.attn1, x, num_views=6, multiview_attention=True)
    c = tmv.JointAttnProcessor()(tb.attn_joint_mid, x)
    e = tmv.XFormersJointAttnProcessor()(tb.attn_joint_mid, x)
    d = max(float((a - b).abs().max()), float((c - e).abs().max()))
    print("xformers-processor vs default-processor, max abs diff:", d)
    assert d < 1e-12
    print("output", tuple(out.shape), "rms", float(out.pow(2).mean().sqrt()))

    arrays = {"out": out.numpy(),
              "names": np.array([n for n, _ in names_shapes]),
              "shapes": np.array([",".join(map(str, s)) for _, s in names_shapes]),
              "keep": np.array(KEEP)}
    for k, v in taps.items():
        v = v[KEEP] if v.shape[0] == 12 else v
        arrays["tap." + k] = v.numpy().astype(np.float32)
        print(f"  tap {k:32s} {tuple(v.shape)} rms {float(v.pow(2).mean().sqrt()):.4f}")
    cfg_items = {k: (list(v) if isinstance(v, tuple) else v) for k, v in CFG.items()}
    arrays["cfg_json"] = np.array(__import__("json").dumps(cfg_items))
    path = os.path.join(ROOT, "tests", "golden", "mv_reference.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(names_shapes), "tensors,",
          sum(int(np.prod(s)) for _, s in names_shapes) // 1000000, "M parameters")


if __name__ == "__main__":
    main()
