"""CPU oracle for the multi-view diffusion path.  TEST INFRASTRUCTURE ONLY.

The reference's mvdiffusion package cannot be imported here (diffusers==0.19.3 and
xformers==0.0.17 are absent and not installable), so this file restates:
  * the reference's OWN attention-processor logic — the K/V regrouping of
    XFormersMVAttnProcessor (mvdiffusion/models/transformer_mv2d.py:783-796) and
    XFormersJointAttnProcessor (:876-883), BasicMVTransformerBlock.forward (:532-625),
    TransformerMV2DModel.forward (:239-374) — literally (einops rearrange/repeat, chunk/cat),
  * xformers.ops.memory_efficient_attention(q,k,v) = softmax(q k^T / sqrt(d)) v on
    (B*H, M, d) tensors (its documented semantics; attn_bias is None on this path),
  * the diffusers 0.19.3 building blocks the UNet is assembled from (Attention projections,
    FeedForward/GEGLU, ResnetBlock2D, Downsample2D/Upsample2D, Timesteps/TimestepEmbedding),
    from their published definitions.
PARITY: the GRAPH this file restates is pinned to the reference's own modules —
tests/golden/mv_reference.npz is produced by running mvdiffusion/models/*.py unmodified (float64,
CPU) over stand-ins for diffusers / xformers (oracle/stubs/), and tests/test_mv_reference_fixture.py
holds UNetRef to it at 1e-9, output and per-block intermediates.  The LEAF OPS of diffusers 0.19.3
/ xformers 0.0.17 stay "parity unpinned" (packages absent, restated from their published
definitions here AND in the stubs); hand-derived KATs for them: tests/test_oracle_mv.py.
Everything is float64 on the CPU (parameters are widened on use, so a full-width 910 M-parameter
state_dict stays in its f16 storage).
"""
import math

import torch
import torch.nn.functional as F
from einops import rearrange, repeat


def memory_efficient_attention(q, k, v):
    """xformers semantics on (B*H, M, d): softmax(q k^T * d^-0.5) v.  Evaluated in slices of the
    batch axis so that the BASELINE shape (96 x 1024 x 6144 scores) stays under 1 GB."""
    out = []
    step = max(1, int(2 ** 27 // max(q.shape[1] * k.shape[1], 1)))
    for b0 in range(0, q.shape[0], step):
        sl = slice(b0, b0 + step)
        s = torch.einsum("bmd,bnd->bmn", q[sl], k[sl]) * (q.shape[-1] ** -0.5)
        out.append(torch.einsum("bmn,bnd->bmd", torch.softmax(s, dim=-1), v[sl]))
    return torch.cat(out, 0)


def head_to_batch_dim(t, heads):          # diffusers Attention.head_to_batch_dim
    b, n, c = t.shape
    return t.reshape(b, n, heads, c // heads).permute(0, 2, 1, 3).reshape(b * heads, n, c // heads)


def batch_to_head_dim(t, heads):
    bh, n, d = t.shape
    return t.reshape(bh // heads, heads, n, d).permute(0, 2, 1, 3).reshape(bh // heads, n, d * heads)


def my_repeat(t, n):                      # transformer_mv2d.py:40-47
    return repeat(t, "b d c -> (b v) d c", v=n)


def mv_attention_core(query, key_raw, value_raw, heads, num_views):
    """transformer_mv2d.py:783-803 between the projections and to_out."""
    key = my_repeat(rearrange(key_raw, "(b t) d c -> b (t d) c", t=num_views), num_views)
    value = my_repeat(rearrange(value_raw, "(b t) d c -> b (t d) c", t=num_views), num_views)
    o = memory_efficient_attention(head_to_batch_dim(query, heads), head_to_batch_dim(key, heads),
                                   head_to_batch_dim(value, heads))
    return batch_to_head_dim(o, heads)


def joint_attention_core(query, key, value, heads):
    """transformer_mv2d.py:876-891."""
    key_0, key_1 = torch.chunk(key, dim=0, chunks=2)
    value_0, value_1 = torch.chunk(value, dim=0, chunks=2)
    key = torch.cat([key_0, key_1], dim=1)
    value = torch.cat([value_0, value_1], dim=1)
    key = torch.cat([key] * 2, dim=0)
    value = torch.cat([value] * 2, dim=0)
    o = memory_efficient_attention(head_to_batch_dim(query, heads), head_to_batch_dim(key, heads),
                                   head_to_batch_dim(value, heads))
    return batch_to_head_dim(o, heads)


# ----------------------------------------------------------------------------------------------
# Functional float64 UNet driven by a state_dict (diffusers 0.19.3 naming), written independently
# of drawingspinup_amd.mv.unet: NCHW tensors, explicit permutes, explicit K/V repeat.
# ----------------------------------------------------------------------------------------------
def timestep_embedding(t, dim, flip_sin_to_cos=True, shift=0, dtype=torch.float64):
    """diffusers get_timestep_embedding.  diffusers evaluates it in float32 whatever the model
    dtype (unet_mv2d_condition.py:867-870 "`Timesteps` ... will always return f32 tensors");
    dtype=torch.float32 reproduces that, the default keeps the exact float64 value."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=dtype) / (half - shift)
    emb = t[:, None].to(dtype) * torch.exp(exponent)[None]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], -1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], -1)
    return emb.double()


class UNetRef:
    def __init__(self, sd, block_out_channels, down_types, up_types, layers_per_block=2, heads=8,
                 groups=32, eps=1e-5, num_views=6, cd_attention_mid=True, temb_dtype=torch.float64,
                 dtype=torch.float64):
        self.sd = dict(sd)               # widened to `dtype` on use (p())
        self.temb_dtype = temb_dtype
        # float64: the oracle the tests compare with.  float32: bench.py's CPU-baseline leg (the
        # arithmetic BASELINE.md section 2 names for the CPU run); never used as a checker.
        self.dtype = dtype
        self.taps = None                 # set to {} to record named intermediates
        self.boc, self.down_types, self.up_types = block_out_channels, down_types, up_types
        self.lpb, self.heads, self.groups, self.eps = layers_per_block, heads, groups, eps
        self.num_views, self.cd_mid = num_views, cd_attention_mid

    def p(self, name):
        return self.sd[name].to(self.dtype)

    def tap(self, name, value):
        if self.taps is not None:
            self.taps[name] = value

    def lin(self, pre, x, bias=True):
        return F.linear(x, self.p(pre + ".weight"), self.p(pre + ".bias") if bias else None)

    def conv(self, pre, x, stride=1, padding=1):
        return F.conv2d(x, self.p(pre + ".weight"), self.p(pre + ".bias"), stride, padding)

    def gn(self, pre, x, eps):
        return F.group_norm(x, self.groups, self.p(pre + ".weight"), self.p(pre + ".bias"), eps)

    def ln(self, pre, x):
        return F.layer_norm(x, (x.shape[-1],), self.p(pre + ".weight"), self.p(pre + ".bias"), 1e-5)

    def resnet(self, pre, x, emb):            # diffusers ResnetBlock2D.forward
        h = F.silu(self.gn(pre + ".norm1", x, self.eps))
        h = self.conv(pre + ".conv1", h)
        h = h + self.lin(pre + ".time_emb_proj", F.silu(emb))[:, :, None, None]
        h = F.silu(self.gn(pre + ".norm2", h, self.eps))
        h = self.conv(pre + ".conv2", h)
        if pre + ".conv_shortcut.weight" in self.sd:
            x = self.conv(pre + ".conv_shortcut", x, padding=0)
        return x + h

    def attn_proj(self, pre, x, ctx=None):
        ctx = x if ctx is None else ctx
        return (self.lin(pre + ".to_q", x, False), self.lin(pre + ".to_k", ctx, False),
                self.lin(pre + ".to_v", ctx, False))

    def block(self, pre, h, ctx):             # BasicMVTransformerBlock.forward
        n1 = self.ln(pre + ".norm1", h)
        self.tap(pre + ".norm1", n1)
        q, k, v = self.attn_proj(pre + ".attn1", n1)
        a1 = self.lin(pre + ".attn1.to_out.0", mv_attention_core(q, k, v, self.heads, self.num_views))
        self.tap(pre + ".attn1", a1)
        h = a1 + h
        if self.cd_mid:
            q, k, v = self.attn_proj(pre + ".attn_joint_mid", self.ln(pre + ".norm_joint_mid", h))
            aj = self.lin(pre + ".attn_joint_mid.to_out.0", joint_attention_core(q, k, v, self.heads))
            self.tap(pre + ".attn_joint_mid", aj)
            h = aj + h
        q, k, v = self.attn_proj(pre + ".attn2", self.ln(pre + ".norm2", h), ctx)
        o = memory_efficient_attention(head_to_batch_dim(q, self.heads),
                                       head_to_batch_dim(k, self.heads),
                                       head_to_batch_dim(v, self.heads))
        a2 = self.lin(pre + ".attn2.to_out.0", batch_to_head_dim(o, self.heads))
        self.tap(pre + ".attn2", a2)
        h = a2 + h
        n = self.ln(pre + ".norm3", h)
        proj = self.lin(pre + ".ff.net.0.proj", n)
        a, g = proj.chunk(2, dim=-1)
        ff = self.lin(pre + ".ff.net.2", a * F.gelu(g))
        self.tap(pre + ".ff", ff)
        self.tap(pre + ".out", ff + h)
        return ff + h

    def transformer(self, pre, x, ctx):       # TransformerMV2DModel.forward
        b, c, hh, ww = x.shape
        res = x
        h = self.gn(pre + ".norm", x, 1e-6)
        h = self.conv(pre + ".proj_in", h, padding=0)
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, -1)
        h = self.block(pre + ".transformer_blocks.0", h, ctx)
        h = h.reshape(b, hh, ww, -1).permute(0, 3, 1, 2)
        return self.conv(pre + ".proj_out", h, padding=0) + res

    def __call__(self, sample, t, ctx, class_labels):
        x, ctx, cl = sample.to(self.dtype), ctx.to(self.dtype), class_labels.to(self.dtype)
        B = x.shape[0]
        temb = timestep_embedding(t.reshape(-1).expand(B), self.boc[0],
                                  dtype=self.temb_dtype).to(self.dtype)
        emb = self.lin("time_embedding.linear_2", F.silu(self.lin("time_embedding.linear_1", temb)))
        self.tap("time_embedding", emb)
        cemb = self.lin("class_embedding.linear_2", F.silu(self.lin("class_embedding.linear_1", cl)))
        self.tap("class_embedding", cemb)
        emb = emb + cemb
        x = self.conv("conv_in", x)
        self.tap("conv_in", x)
        skips = [x]
        for i, typ in enumerate(self.down_types):
            for j in range(self.lpb):
                x = self.resnet(f"down_blocks.{i}.resnets.{j}", x, emb)
                self.tap(f"down_blocks.{i}.resnets.{j}", x)
                if typ.startswith("CrossAttn"):
                    x = self.transformer(f"down_blocks.{i}.attentions.{j}", x, ctx)
                    self.tap(f"down_blocks.{i}.attentions.{j}", x)
                skips.append(x)
            if i != len(self.down_types) - 1:
                x = self.conv(f"down_blocks.{i}.downsamplers.0.conv", x, stride=2)
                skips.append(x)
            self.tap(f"down_blocks.{i}", x)
        x = self.resnet("mid_block.resnets.0", x, emb)
        x = self.transformer("mid_block.attentions.0", x, ctx)
        x = self.resnet("mid_block.resnets.1", x, emb)
        self.tap("mid_block", x)
        for i, typ in enumerate(self.up_types):
            for j in range(self.lpb + 1):
                x = torch.cat([x, skips.pop()], 1)
                x = self.resnet(f"up_blocks.{i}.resnets.{j}", x, emb)
                if typ.startswith("CrossAttn"):
                    x = self.transformer(f"up_blocks.{i}.attentions.{j}", x, ctx)
            if i != len(self.up_types) - 1:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                x = self.conv(f"up_blocks.{i}.upsamplers.0.conv", x)
            self.tap(f"up_blocks.{i}", x)
        x = F.silu(self.gn("conv_norm_out", x, self.eps))
        return self.conv("conv_out", x)


# ----------------------------------------------------------------------------------------------
# diffusers 0.19.3 DDIMScheduler (schedulers/scheduling_ddim.py: __init__, set_timesteps with
# timestep_spacing="leading", step) and the denoising loop of
# mvdiffusion/pipelines/pipeline_mvdiffusion_image.py:463-486, in float64 with injected noise.
# ----------------------------------------------------------------------------------------------
def ddim_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """beta_schedule="scaled_linear": linspace in sqrt(beta), float32 as diffusers builds it."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                           dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0).double()


def ddim_timesteps(num_inference_steps, num_train_timesteps=1000, steps_offset=1):
    ratio = num_train_timesteps // num_inference_steps
    return [int(round(i * ratio)) + steps_offset for i in range(num_inference_steps)][::-1]


def ddim_step(model_output, t, sample, num_inference_steps, eta, variance_noise, acp=None,
              num_train_timesteps=1000, set_alpha_to_one=False):
    """One DDIMScheduler.step (epsilon prediction, clip_sample False, use_clipped False): formulas
    (12) and (16) of Song et al. as diffusers writes them."""
    acp = ddim_alphas_cumprod(num_train_timesteps) if acp is None else acp
    prev_t = t - num_train_timesteps // num_inference_steps
    a_t = acp[t]
    a_prev = acp[prev_t] if prev_t >= 0 else (torch.tensor(1.0, dtype=torch.float64)
                                              if set_alpha_to_one else acp[0])
    b_t = 1 - a_t
    x, eps = sample.double(), model_output.double()
    x0 = (x - b_t ** 0.5 * eps) / a_t ** 0.5
    variance = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
    std = eta * variance ** 0.5
    prev = a_prev ** 0.5 * x0 + (1 - a_prev - std ** 2) ** 0.5 * eps
    if eta > 0:
        prev = prev + std * variance_noise.double()
    return prev


def denoise_loop(unet, latents, image_latents, image_embeddings, camera_embeddings,
                 num_inference_steps, step_noise, eta=1.0, run_steps=None, round_dtype=None):
    """pipeline_mvdiffusion_image.py:463-486 without classifier-free guidance (mv.py:81):
    cat(latents, image_latents) -> UNet -> scheduler.step.  `round_dtype` (torch.float16) rounds
    the latents after every step as the reference's f16 pipeline does (prev_sample keeps the
    sample dtype); returns the list of latents after each step."""
    acp = ddim_alphas_cumprod()
    out = []
    lat = latents.double()
    for i, t in enumerate(ddim_timesteps(num_inference_steps)[:run_steps]):
        model_in = torch.cat([lat, image_latents.double()], 1)
        eps = unet(model_in, torch.tensor([t]), image_embeddings, camera_embeddings)
        if round_dtype is not None:
            eps = eps.to(round_dtype).double()
        lat = ddim_step(eps, t, lat, num_inference_steps, eta, step_noise[i], acp)
        if round_dtype is not None:
            lat = lat.to(round_dtype).double()
        out.append(lat)
    return out
