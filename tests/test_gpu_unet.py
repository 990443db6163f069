"""gfx950 UNet (NHWC f16, HIP conv/attention/norm kernels) vs the float64 functional oracle."""
import pytest
import torch

from drawingspinup_amd import ops
from drawingspinup_amd.mv.unet import UNetMV2DConditionModel
from oracle import mv_ref as mr

pytestmark = pytest.mark.gpu

SMALL = dict(block_out_channels=(320, 640, 640), layers_per_block=1,
             down_block_types=("CrossAttnDownBlockMV2D", "CrossAttnDownBlockMV2D", "DownBlock2D"),
             up_block_types=("UpBlock2D", "CrossAttnUpBlockMV2D", "CrossAttnUpBlockMV2D"))


def _init(model, seed):
    g = torch.Generator().manual_seed(seed)
    for n, p in model.named_parameters():
        if p.dim() > 1:
            fan_in = p[0].numel()
            p.data = torch.randn(p.shape, generator=g) * fan_in ** -0.5
        elif "norm" in n and n.endswith("weight"):
            p.data = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        else:
            p.data = 0.05 * torch.randn(p.shape, generator=g)
    # the zero-initialised joint to_out (transformer_mv2d.py:499,516) would hide that branch
    return model


def test_norm_kernels(dev):
    g = torch.Generator().manual_seed(0)
    # every channels-per-group of the UNet (10 ... 80) and the VAE's 16: the 16-byte super-group
    # kernel where a super-group fits one workgroup's registers, else the one-launch half2 kernel
    # ((3,5,7)) or the statistics + apply pair ((1,5,7), (1,64,64) at 320: 20480 chunks);
    # DSU_GN_SUPER=0 sends everything down the older paths (tools/visit_r3_36.sh runs both)
    for C, shp in [(320, (3, 5, 7)), (640, (3, 5, 7)), (1280, (3, 5, 7)), (1920, (3, 5, 7)),
                   (2560, (3, 5, 7)), (640, (1, 5, 7)), (320, (1, 64, 64)), (960, (12, 32, 32)),
                   (320, (2, 32, 32)), (640, (2, 16, 16)), (1280, (2, 8, 8)), (512, (1, 64, 64)),
                   (256, (1, 70, 50))]:
        x = torch.randn(*shp, C, generator=g).half()
        w, b = (1 + 0.1 * torch.randn(C, generator=g)).half(), (0.1 * torch.randn(C, generator=g)).half()
        for silu in (False, True):
            ref = torch.nn.functional.group_norm(x.float().permute(0, 3, 1, 2), 32, w.float(),
                                                 b.float(), 1e-5)
            ref = torch.nn.functional.silu(ref) if silu else ref
            got = ops.groupnorm_nhwc_f16(x.to(dev), w.to(dev), b.to(dev), 32, 1e-5, silu)
            torch.testing.assert_close(got.cpu().float().permute(0, 3, 1, 2), ref, rtol=2e-3, atol=2e-3)
    for C in (320, 640, 1280):
        x = torch.randn(37, C, generator=g).half()
        w, b = (1 + 0.1 * torch.randn(C, generator=g)).half(), (0.1 * torch.randn(C, generator=g)).half()
        ref = torch.nn.functional.layer_norm(x.float(), (C,), w.float(), b.float(), 1e-5)
        torch.testing.assert_close(ops.layernorm_f16(x.to(dev), w.to(dev), b.to(dev)).cpu().float(),
                                   ref, rtol=2e-3, atol=2e-3)
    h = torch.randn(19, 2 * 1280, generator=g).half()
    a, gt = h.float().chunk(2, -1)
    torch.testing.assert_close(ops.geglu_f16(h.to(dev)).cpu().float(),
                               a * torch.nn.functional.gelu(gt), rtol=2e-3, atol=2e-3)


def test_unet_small_vs_oracle(dev):
    torch.manual_seed(0)
    model = _init(UNetMV2DConditionModel(**SMALL), 1).half()
    sd = {k: v.clone() for k, v in model.state_dict().items()}     # f16 values, shared by both
    g = torch.Generator().manual_seed(2)
    sample = torch.randn(12, 8, 8, 8, generator=g).half()
    ctx = torch.randn(12, 1, 768, generator=g).half()
    cl = torch.randn(12, 10, generator=g).half()
    t = torch.tensor([487])
    ref = mr.UNetRef(sd, SMALL["block_out_channels"], SMALL["down_block_types"],
                     SMALL["up_block_types"], layers_per_block=1)(sample, t, ctx, cl)
    out = model.to(dev)(sample.to(dev), t.to(dev), ctx.to(dev), cl.to(dev)).cpu().double()
    rel = (out - ref).norm() / ref.norm()
    print("unet rel-L2", float(rel))
    assert out.shape == (12, 4, 8, 8)
    assert float(rel) < 5e-3           # f16 activations through ~40 layers vs float64


FULL = dict(block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
            down_block_types=("CrossAttnDownBlockMV2D", "CrossAttnDownBlockMV2D",
                              "CrossAttnDownBlockMV2D", "DownBlock2D"),
            up_block_types=("UpBlock2D", "CrossAttnUpBlockMV2D", "CrossAttnUpBlockMV2D",
                            "CrossAttnUpBlockMV2D"))
_FULL_CACHE = {}


def _full_model(dev):
    """BASELINE config 2 architecture (910 M parameters), one instance for the module's tests."""
    if "m" not in _FULL_CACHE:
        torch.manual_seed(0)
        model = _init(UNetMV2DConditionModel(**FULL), 11).half()
        sd = {k: v.clone() for k, v in model.state_dict().items()}      # f16, shared with the oracle
        _FULL_CACHE["m"] = (model.to(dev).eval(), sd)
    return _FULL_CACHE["m"]


def _full_ref(sd):
    return mr.UNetRef(sd, FULL["block_out_channels"], FULL["down_block_types"],
                      FULL["up_block_types"], layers_per_block=2)


@pytest.mark.parametrize("side", [16, 32])
def test_unet_full_width_vs_oracle(dev, side):
    """SURVEY 8(d) tolerance on the SHIPPED architecture (320/640/1280/1280, two layers per block,
    16 multi-view transformer blocks, B = 12): rel-L2 <= 2e-3 against the float64 restatement.
    side = 32 is BASELINE config 2's exact input shape (12, 8, 32, 32); side = 16 is the smallest
    latent the attention kernel takes (its deepest level then has 2x2 = 4 tokens per view; the
    kernel stages V^T four keys at a time and refuses shorter segments with DSU_EUNSUP)."""
    model, sd = _full_model(dev)
    g = torch.Generator().manual_seed(20 + side)
    sample = torch.randn(12, 8, side, side, generator=g).half()
    ctx = torch.randn(12, 1, 768, generator=g).half()
    cl = torch.randn(12, 10, generator=g).half()
    t = torch.tensor([487])
    with torch.no_grad():
        out = model(sample.to(dev), t.to(dev), ctx.to(dev), cl.to(dev)).cpu().double()
    ref = _full_ref(sd)(sample, t, ctx, cl)
    rel = float((out - ref).norm() / ref.norm())
    print(f"full-width unet {side}x{side} rel-L2 {rel:.2e}")
    assert out.shape == (12, 4, side, side)
    assert rel < 2e-3


def test_ddim_steps_vs_oracle_loop(dev):
    """pipeline_mvdiffusion_image.py:463-486 with injected latents and per-step noise, full-width
    UNet at 16x16 latents: latents after ONE step rel-L2 <= 2e-3, after four steps
    <= 1e-2 (reported).  The oracle loop rounds the model output and the latents to f16 after
    every step, as the reference's f16 pipeline does."""
    from drawingspinup_amd.mv.pipeline import AutoencoderKL, MVDiffusionImagePipeline
    model, sd = _full_model(dev)
    vae = AutoencoderKL().half().to(dev).eval()
    pipe = MVDiffusionImagePipeline(model, vae, None)
    g = torch.Generator().manual_seed(31)
    B, steps, run = 12, 75, 4
    emb = (torch.randn(B, 1, 768, generator=g) * 0.5).half()
    img_lat = torch.randn(B, 4, 16, 16, generator=g).half()
    pipe._encode_image = lambda images: (emb.to(dev), img_lat.to(dev))
    lat0 = torch.randn(B, 4, 16, 16, generator=g).half()
    noise = torch.randn(steps, B, 4, 16, 16, generator=g).half()
    got = []
    sched = pipe.scheduler
    orig_set = sched.set_timesteps

    def first_steps(n, device=None):                   # the 75-step schedule, first `run` steps
        orig_set(n, device=device)
        sched.timesteps = sched.timesteps[:run]
    sched.set_timesteps = first_steps
    pipe(torch.zeros(B, 3, 128, 128), height=128, width=128, num_inference_steps=steps,
         latents=lat0.clone(), step_noise=noise, output_type="latent", eta=1.0,
         callback=lambda i, t, lat: got.append(lat.float().cpu().double()))
    cam = pipe.prepare_camera_embedding(
        __import__("drawingspinup_amd.mv.pipeline", fromlist=["x"]).DEFAULT_CAMERA_EMBEDDING).cpu()
    ref = mr.denoise_loop(_full_ref(sd), lat0, img_lat, emb, cam, steps, noise, eta=1.0,
                          run_steps=run, round_dtype=torch.float16)
    rels = [float((a - b).norm() / b.norm()) for a, b in zip(got, ref)]
    print("ddim rel-L2 per step", ["%.2e" % r for r in rels])
    assert len(got) == run
    assert rels[0] < 2e-3
    assert rels[-1] < 1e-2


def test_ddim_75_steps_vs_oracle_fixture(dev):
    """The WHOLE 75-step loop (eta = 1, injected latents and per-step noise) on the shipped UNet at
    16x16 latents against the float64 oracle's latents, computed offline by
    tests/golden/make_ddim75_golden.py (same seeds -> same weights and inputs; the fixture carries a
    checksum of the weights).  SURVEY.md 8(d): "after 75 steps: report, expect <= 2e-2"."""
    import os
    import numpy as np
    from drawingspinup_amd.mv.pipeline import AutoencoderKL, MVDiffusionImagePipeline
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ddim75_reference.npz"))
    model, sd = _full_model(dev)
    checksum = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(checksum - float(gold["weights_abs_sum"])) < 1e-9 * checksum
    pipe = MVDiffusionImagePipeline(model, None, None)
    g = torch.Generator().manual_seed(31)
    B, steps = 12, 75
    emb = (torch.randn(B, 1, 768, generator=g) * 0.5).half()
    img_lat = torch.randn(B, 4, 16, 16, generator=g).half()
    pipe._encode_image = lambda images: (emb.to(dev), img_lat.to(dev))
    lat0 = torch.randn(B, 4, 16, 16, generator=g).half()
    noise = torch.randn(steps, B, 4, 16, 16, generator=g).half()
    got = []
    pipe(torch.zeros(B, 3, 128, 128), height=128, width=128, num_inference_steps=steps,
         latents=lat0.clone(), step_noise=noise, output_type="latent", eta=1.0,
         callback=lambda i, t, lat: got.append(lat.float().cpu().double()))
    assert len(got) == steps
    rels = {}
    for k in gold["steps"].tolist():
        ref = torch.from_numpy(gold["lat_%d" % k]).double()
        rels[k] = float((got[k - 1] - ref).norm() / ref.norm())
    print("ddim rel-L2 after k steps:", {k: "%.2e" % r for k, r in rels.items()})
    out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "ddim75_rel_l2.txt"), "w") as f:
            f.write("rel-L2 of the latents vs the float64 oracle after k of 75 DDIM steps "
                    "(16x16 latents, B=12, eta=1, injected noise)\n")
            f.writelines("%d %.4e\n" % kv for kv in rels.items())
    assert rels[1] < 2e-3
    assert rels[75] < 2e-2


def test_vae_vs_torch_reference(dev):
    """AutoencoderKL encode(mode)/decode on the HIP conv/norm kernels vs the same weights run
    through plain torch f32 ops on the CPU (diffusers AutoencoderKL structure)."""
    import torch.nn.functional as F
    from drawingspinup_amd.mv.pipeline import AutoencoderKL
    torch.manual_seed(0)
    vae = _init(AutoencoderKL(), 3).half()
    sd = {k: v.float() for k, v in vae.state_dict().items()}

    def gn(p, x, silu=True):
        y = F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)
        return F.silu(y) if silu else y

    def conv(p, x, stride=1, pad=1):
        return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride, pad)

    def res(p, x):
        h = conv(p + ".conv1", gn(p + ".norm1", x))
        h = conv(p + ".conv2", gn(p + ".norm2", h))
        if p + ".conv_shortcut.weight" in sd:
            x = conv(p + ".conv_shortcut", x, pad=0)
        return x + h

    def attn(p, x):
        B, C, H, W = x.shape
        h = gn(p + ".group_norm", x, False).flatten(2).transpose(1, 2)
        q, k, v = [F.linear(h, sd[f"{p}.to_{n}.weight"], sd[f"{p}.to_{n}.bias"]) for n in "qkv"]
        a = torch.softmax(q @ k.transpose(1, 2) * C ** -0.5, -1) @ v
        o = F.linear(a, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
        return x + o.transpose(1, 2).reshape(B, C, H, W)

    def mid(p, x):
        return res(p + ".resnets.1", attn(p + ".attentions.0", res(p + ".resnets.0", x)))

    g = torch.Generator().manual_seed(1)
    img = (torch.rand(1, 3, 64, 64, generator=g) * 2 - 1).half()
    x = conv("encoder.conv_in", img.float())
    for i in range(4):
        for j in range(2):
            x = res(f"encoder.down_blocks.{i}.resnets.{j}", x)
        if i < 3:
            x = conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", F.pad(x, (0, 1, 0, 1)), 2, 0)
    x = conv("encoder.conv_out", gn("encoder.conv_norm_out", mid("encoder.mid_block", x)))
    ref_lat = conv("quant_conv", x, pad=0)[:, :4]
    lat = vae.to(dev).encode_mode(img.to(dev)).cpu().float()
    rel = (lat - ref_lat).norm() / ref_lat.norm()
    assert lat.shape == (1, 4, 8, 8) and float(rel) < 1e-2, float(rel)

    z = torch.randn(1, 4, 8, 8, generator=g).half()
    x = conv("decoder.conv_in", conv("post_quant_conv", z.float(), pad=0))
    x = mid("decoder.mid_block", x)
    for i in range(4):
        for j in range(3):
            x = res(f"decoder.up_blocks.{i}.resnets.{j}", x)
        if i < 3:
            x = conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", F.interpolate(x, scale_factor=2.0, mode="nearest"))
    ref_img = conv("decoder.conv_out", gn("decoder.conv_norm_out", x))
    out = vae.decode(z.to(dev)).cpu().float()
    rel = (out - ref_img).norm() / ref_img.norm()
    assert out.shape == (1, 3, 64, 64) and float(rel) < 1e-2, float(rel)


def test_pipeline_graph_replay_matches_eager(dev):
    """The denoising loop with the UNet step replayed from a captured HIP graph returns the same
    latents as the eager loop (same kernels, same order; injected noise)."""
    from drawingspinup_amd.mv.pipeline import AutoencoderKL, MVDiffusionImagePipeline
    torch.manual_seed(0)
    unet = _init(UNetMV2DConditionModel(**SMALL), 5).half().to(dev).eval()
    vae = AutoencoderKL().half().to(dev).eval()
    pipe = MVDiffusionImagePipeline(unet, vae, None)
    g = torch.Generator().manual_seed(6)
    B, steps = 12, 4
    emb = (torch.randn(B, 1, 768, generator=g) * 0.5).half().to(dev)
    img_lat = torch.randn(B, 4, 8, 8, generator=g).half().to(dev)
    pipe._encode_image = lambda images: (emb, img_lat)
    lat0 = torch.randn(B, 4, 8, 8, generator=g).half()
    noise = torch.randn(steps, B, 4, 8, 8, generator=g).half()
    image = torch.zeros(B, 3, 64, 64)
    outs = []
    for use_graph in (False, True):
        pipe.use_graph, pipe._graph = use_graph, None
        outs.append(pipe(image, height=64, width=64, num_inference_steps=steps, latents=lat0.clone(),
                         step_noise=noise, output_type="latent").float().cpu())
    assert pipe._graph is not None and pipe._graph["graph"] is not None
    assert torch.isfinite(outs[0]).all()
    torch.testing.assert_close(outs[1], outs[0], rtol=0, atol=0)


def test_unet_vs_reference_module_fixture(dev):
    """The HIP UNet against the REFERENCE's own UNetMV2DConditionModel: tests/golden/
    mv_reference.npz is mvdiffusion/models/*.py run unmodified (float64, CPU, over oracle/stubs;
    tests/golden/make_mv_reference_golden.py) at reduced width — 80/160/320/320 channels, 2 heads of
    40/80/160/160, 8 norm groups, (12, 8, 16, 16) input, parameters regenerated from their names.
    Output rel-L2 < 5e-3 (f16 activations and weights through ~60 layers vs float64), and the
    per-block intermediates the fixture recorded (rows `keep` of the batch) < 1e-2 each."""
    import json
    import os
    import numpy as np
    from oracle import mv_weights
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mv_reference.npz"))
    cfg = json.loads(str(z["cfg_json"]))
    names_shapes = [(str(n), tuple(int(v) for v in str(s).split(",")) if str(s) else ())
                    for n, s in zip(z["names"], z["shapes"])]
    model = UNetMV2DConditionModel(
        sample_size=cfg["sample_size"], in_channels=cfg["in_channels"],
        out_channels=cfg["out_channels"], block_out_channels=tuple(cfg["block_out_channels"]),
        layers_per_block=cfg["layers_per_block"], cross_attention_dim=cfg["cross_attention_dim"],
        attention_head_dim=cfg["attention_head_dim"], norm_num_groups=cfg["norm_num_groups"],
        projection_class_embeddings_input_dim=cfg["projection_class_embeddings_input_dim"],
        num_views=cfg["num_views"], cd_attention_mid=cfg["cd_attention_mid"],
        down_block_types=tuple(cfg["down_block_types"]), up_block_types=tuple(cfg["up_block_types"]))
    model.load_state_dict({k: v.float() for k, v in mv_weights.synth_state_dict(names_shapes).items()},
                          strict=True)
    model = model.half().to(dev).eval()

    taps = {}

    def tap(name, nhwc=True):
        def hook(_m, _i, out):
            o = out.detach().float().cpu().double()
            taps[name] = o.permute(0, 3, 1, 2) if (nhwc and o.dim() == 4) else o
        return hook

    for i, b in enumerate(model.down_blocks):
        last = b.downsamplers[0] if hasattr(b, "downsamplers") else b.resnets[-1]
        last.register_forward_hook(tap(f"down_blocks.{i}"))
    model.mid_block.resnets[1].register_forward_hook(tap("mid_block"))
    for i, b in enumerate(model.up_blocks):
        last = b.upsamplers[0] if hasattr(b, "upsamplers") else (
            b.attentions[-1] if hasattr(b, "attentions") else b.resnets[-1])
        last.register_forward_hook(tap(f"up_blocks.{i}"))
    model.down_blocks[0].resnets[0].register_forward_hook(tap("down_blocks.0.resnets.0"))
    model.down_blocks[0].attentions[0].register_forward_hook(tap("down_blocks.0.attentions.0"))
    model.down_blocks[0].attentions[0].transformer_blocks[0].register_forward_hook(tap("tb.out"))

    sample = mv_weights.det_tensor("in.sample", (12, 8, 16, 16), 1.5).half()
    ctx = mv_weights.det_tensor("in.ctx", (12, 1, 768), 1.0).half()
    cam = mv_weights.det_tensor("in.cam", (12, 5), 3.0)
    cl = torch.cat([torch.sin(cam), torch.cos(cam)], -1).half()
    t = torch.tensor([487])
    with torch.no_grad():
        out = model(sample.to(dev), t.to(dev), ctx.to(dev), cl.to(dev)).cpu().double()
    want = torch.from_numpy(z["out"])
    rel = float((out - want).norm() / want.norm())
    print(f"HIP unet vs reference-module fixture: rel-L2 {rel:.2e}")
    keep = z["keep"]
    worst = {}
    for name, got in taps.items():
        w = torch.from_numpy(z["tap." + name]).double()
        g = got[keep].reshape(w.shape)
        worst[name] = float((g - w).norm() / w.norm())
    print("intermediates rel-L2:", {k: "%.1e" % v for k, v in worst.items()})
    assert rel < 5e-3
    assert len(worst) >= 11 and max(worst.values()) < 1e-2


def test_pipeline_vs_reference_pipeline_fixture(dev):
    """drawingspinup_amd.mv.pipeline.MVDiffusionImagePipeline (HIP UNet, f16) against the
    REFERENCE's own MVDiffusionImagePipeline.__call__ driving its own UNet in float64
    (tests/golden/mv_pipeline_reference.npz, make_mv_pipeline_golden.py): same f16 input batch,
    camera embeddings, injected initial latents and per-step variance noise; 3 DDIM steps with
    eta = 1, decode, denormalise.  Latents after each step rel-L2 < 5e-3; images mean |d| < 3e-3 and
    max |d| < 8e-2 (the linear stand-in decoder multiplies a latent error by 1 / scaling_factor =
    5.5 and by its 4 -> 192 channel map before the clamp to [0, 1])."""
    import json
    import os
    import numpy as np
    from drawingspinup_amd.mv.pipeline import MVDiffusionImagePipeline
    from oracle import mv_weights
    from oracle.mv_pipeline_aux import (LinearClip, LinearVAE, aux_state, camera_embeddings,
                                        det_noise, input_image)
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mv_pipeline_reference.npz"))
    cfg = json.loads(str(z["cfg_json"]))
    names_shapes = [(str(n), tuple(int(v) for v in str(s).split(",")) if str(s) else ())
                    for n, s in zip(z["names"], z["shapes"])]
    unet = UNetMV2DConditionModel(
        sample_size=cfg["sample_size"], in_channels=cfg["in_channels"],
        out_channels=cfg["out_channels"], block_out_channels=tuple(cfg["block_out_channels"]),
        layers_per_block=cfg["layers_per_block"], cross_attention_dim=cfg["cross_attention_dim"],
        attention_head_dim=cfg["attention_head_dim"], norm_num_groups=cfg["norm_num_groups"],
        projection_class_embeddings_input_dim=cfg["projection_class_embeddings_input_dim"],
        num_views=cfg["num_views"], cd_attention_mid=cfg["cd_attention_mid"],
        down_block_types=tuple(cfg["down_block_types"]), up_block_types=tuple(cfg["up_block_types"]))
    unet.load_state_dict({k: v.float() for k, v in mv_weights.synth_state_dict(names_shapes).items()},
                         strict=True)
    unet = unet.half().to(dev).eval()
    vae = aux_state(LinearVAE().double().eval(), "aux.vae.").half().to(dev)
    clip = aux_state(LinearClip().double().eval(), "aux.clip.").half().to(dev)
    pipe = MVDiffusionImagePipeline(unet, vae, clip)
    steps = int(z["steps"])
    imgs = input_image()[None].expand(12, -1, -1, -1).contiguous()
    noise = torch.stack([det_noise("draw.%d" % (i + 1), (12, 4, 32, 32)) for i in range(steps)]).half()
    got = []
    out = pipe(imgs.to(dev), camera_embeddings().to(dev), num_inference_steps=steps, guidance_scale=1.0,
               eta=1.0, latents=det_noise("draw.0", (12, 4, 32, 32)).half(), step_noise=noise,
               output_type="pt", callback=lambda i, t, lat: got.append(lat.float().cpu().double()))
    assert [int(t) for t in pipe.scheduler.timesteps] == z["timesteps"].tolist()
    rels = []
    for i, lat in enumerate(got):
        want = torch.from_numpy(z["lat_%d" % (i + 1)]).double()
        rels.append(float((lat - want).norm() / want.norm()))
    want_img = torch.from_numpy(z["out"].astype(np.float32))
    d = (out.float().cpu()[z["keep"]] - want_img).abs()
    print("pipeline vs reference pipeline: latents rel-L2", ["%.1e" % r for r in rels],
          "image max|d| %.2e mean|d| %.2e" % (float(d.max()), float(d.mean())))
    assert out.shape == (12, 3, 256, 256) and len(got) == steps
    assert max(rels) < 5e-3
    assert float(d.mean()) < 3e-3 and float(d.max()) < 8e-2
