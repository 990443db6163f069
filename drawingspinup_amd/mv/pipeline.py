"""MVDiffusionImagePipeline on the gfx950 kernels.

Mirrors 2_charactor_reconstructor/mvdiffusion/pipelines/pipeline_mvdiffusion_image.py
(__call__ :299-508, _encode_image :150-182, prepare_latents :254-269,
prepare_camera_embedding :271-296) and the diffusers==0.19.3 pieces it drives
(DDIMScheduler, AutoencoderKL, VaeImageProcessor.postprocess), restated from their published
definitions.  Same call signature for the arguments mv.py uses (mv.py:79-86):
    pipeline(imgs_in, camera_embeddings, generator=..., guidance_scale=1.0, output_type='pt',
             num_images_per_prompt=1, eta=1.0, num_inference_steps=75)
plus `latents=` / `step_noise=` to inject the random draws (tests; RNG streams differ between
CUDA and HIP).
"""
import math

import numpy as np

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from . import preprocess as PP
from .unet import (Downsample2D, Upsample2D, conv_nhwc, group_norm)


def _conv_small_cin(conv, x_nhwc):
    """3x3 conv whose input has fewer than 8 channels (VAE conv_in: 3 or 4): zero-pad channels
    and weights to 8 so the HIP kernel's 16-byte channel groups apply (cached per weight)."""
    w = conv.weight
    c = getattr(conv, "_dsu_okc8", None)
    if c is None or c[0] != w._version or c[1].device != w.device:
        wp = F.pad(w.detach(), (0, 0, 0, 0, 0, 8 - w.shape[1]))
        conv._dsu_okc8 = (w._version, ops.conv_weight_okc(wp))
        c = conv._dsu_okc8
    xp = F.pad(x_nhwc, (0, 8 - x_nhwc.shape[-1])).contiguous()
    return ops.conv2d_nhwc_f16(xp, c[1], conv.bias, 3, 1, 1)


def _conv1x1_nhwc(conv, x_nhwc):
    w = conv.weight.view(conv.out_channels, conv.in_channels)
    if not (x_nhwc.is_cuda and x_nhwc.dtype == torch.float16):
        raise RuntimeError("the gfx950 VAE takes f16 device tensors (no library / CPU fallback)")
    shp = x_nhwc.shape
    x2 = x_nhwc.reshape(-1, shp[-1])
    if shp[-1] % 8:                       # quant / post_quant convs: 8 -> 8 and 4 -> 4 channels
        pad = 8 - shp[-1] % 8             # zero columns: 16-byte rows for the GEMM's loads
        x2, w = F.pad(x2, (0, pad)), F.pad(w, (0, pad))
    return ops.linear_f16(x2.contiguous(), w.contiguous(), conv.bias).view(*shp[:-1], -1)


# ----------------------------------------------------------------------------------- scheduler
class DDIMScheduler:
    """diffusers 0.19.3 DDIMScheduler with the Stable-Diffusion-1.x config the Wonder3D
    checkpoint ships (scaled_linear betas 0.00085..0.012, 1000 train steps, steps_offset 1,
    clip_sample False, set_alpha_to_one False, epsilon prediction, 'leading' spacing)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 steps_offset=1, set_alpha_to_one=False, prediction_type="epsilon"):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                               dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps, self.steps_offset = num_train_timesteps, steps_offset
        self.prediction_type = prediction_type
        self.init_noise_sigma = 1.0
        self.order = 1
        self.timesteps = None

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        step_ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps_host = [int(v) for v in ts + self.steps_offset]   # no device read-back per step
        self.timesteps = torch.from_numpy(ts + self.steps_offset).to(device)

    def scale_model_input(self, sample, t):
        return sample

    def step(self, model_output, timestep, sample, eta=0.0, generator=None, variance_noise=None):
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        mo, x = model_output.float(), sample.float()
        if self.prediction_type == "epsilon":
            x0 = (x - b_t ** 0.5 * mo) / a_t ** 0.5
            eps = mo
        elif self.prediction_type == "v_prediction":
            x0 = a_t ** 0.5 * x - b_t ** 0.5 * mo
            eps = a_t ** 0.5 * mo + b_t ** 0.5 * x
        else:
            raise ValueError(self.prediction_type)
        variance = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
        std = eta * variance ** 0.5
        direction = (1 - a_prev - std ** 2) ** 0.5 * eps
        prev = a_prev ** 0.5 * x0 + direction
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator,
                                             device=model_output.device, dtype=model_output.dtype)
            prev = prev + std * variance_noise.float()
        return prev.to(sample.dtype)


# ----------------------------------------------------------------------------------- VAE
class VaeResnet(nn.Module):
    def __init__(self, cin, cout, groups=32, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1, 1, 0) if cin != cout else None

    def forward(self, x):
        h = conv_nhwc(self.conv1, group_norm(self.norm1, x, silu=True))
        h = group_norm(self.norm2, h, silu=True)
        sc = x if self.conv_shortcut is None else conv_nhwc(self.conv_shortcut, x)
        return conv_nhwc(self.conv2, h, residual=sc)


class VaeAttention(nn.Module):
    """single-head spatial self-attention of the VAE mid block (run once per encode / decode; one
    head of d = 512): projections, Q K^T and P V on the library's own f16 GEMM (the `to_v`
    projection is produced transposed, so P V is a plain `x W^T` product as well); softmax in f32
    as diffusers' AttnProcessor upcasts it."""

    def __init__(self, ch, groups=32, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(ch, ch), nn.Linear(ch, ch), nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Dropout(0.0)])

    def forward(self, x):
        B, H, W, C = x.shape
        h = group_norm(self.group_norm, x).view(B, H * W, C)
        q = ops.linear_f16(h, self.to_q.weight, self.to_q.bias)
        k = ops.linear_f16(h, self.to_k.weight, self.to_k.bias)
        # V^T (B, C, N): the bias is per channel = per ROW of the transposed output
        vt = ops.linear_f16(h, self.to_v.weight, transposed_tokens=H * W) \
            + self.to_v.bias.view(1, C, 1)
        o = torch.empty_like(q)
        for b in range(B):
            s_ = ops.linear_f16(q[b], k[b].contiguous())                       # (N, N) = Q K^T
            a = torch.softmax(s_.float() * C ** -0.5, -1).to(q.dtype)
            o[b] = ops.linear_f16(a, vt[b].contiguous())                       # (N, C) = P V
        o = ops.linear_f16(o, self.to_out[0].weight, self.to_out[0].bias,
                           residual=x.reshape(B, H * W, C))
        return o.view(B, H, W, C)


class _B(nn.Module):
    pass


class VaeMid(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.attentions = nn.ModuleList([VaeAttention(ch)])
        self.resnets = nn.ModuleList([VaeResnet(ch, ch), VaeResnet(ch, ch)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class VaeEncoder(nn.Module):
    def __init__(self, cin=3, chans=(128, 256, 512, 512), latent=4):
        super().__init__()
        self.conv_in = nn.Conv2d(cin, chans[0], 3, 1, 1)
        self.down_blocks = nn.ModuleList()
        c = chans[0]
        for i, co in enumerate(chans):
            b = _B()
            b.resnets = nn.ModuleList([VaeResnet(c if j == 0 else co, co) for j in range(2)])
            if i != len(chans) - 1:
                b.downsamplers = nn.ModuleList([Downsample2D(co)])
                b.downsamplers[0].conv.padding = (0, 0)
            self.down_blocks.append(b)
            c = co
        self.mid_block = VaeMid(c)
        self.conv_norm_out = nn.GroupNorm(32, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, 2 * latent, 3, padding=1)

    def forward(self, x_nchw):
        x = _conv_small_cin(self.conv_in, x_nchw.permute(0, 2, 3, 1))
        for b in self.down_blocks:
            for r in b.resnets:
                x = r(x)
            if hasattr(b, "downsamplers"):
                # diffusers Downsample2D(padding=0): F.pad(x, (0,1,0,1)) then stride-2 conv
                xp = F.pad(x, (0, 0, 0, 1, 0, 1)).contiguous()           # NHWC: pad W and H by (0,1)
                x = conv_nhwc(b.downsamplers[0].conv, xp)                  # stride 2, padding 0
        x = self.mid_block(x)
        x = group_norm(self.conv_norm_out, x, silu=True)
        return conv_nhwc(self.conv_out, x)


class VaeDecoder(nn.Module):
    def __init__(self, cout=3, chans=(128, 256, 512, 512), latent=4):
        super().__init__()
        rev = list(reversed(chans))
        self.conv_in = nn.Conv2d(latent, rev[0], 3, 1, 1)
        self.mid_block = VaeMid(rev[0])
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            b = _B()
            b.resnets = nn.ModuleList([VaeResnet(c if j == 0 else co, co) for j in range(3)])
            if i != len(rev) - 1:
                b.upsamplers = nn.ModuleList([Upsample2D(co)])
            self.up_blocks.append(b)
            c = co
        self.conv_norm_out = nn.GroupNorm(32, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, cout, 3, padding=1)

    def forward(self, z_nchw):
        x = _conv_small_cin(self.conv_in, z_nchw)            # z arrives NHWC, 4 channels
        x = self.mid_block(x)
        for b in self.up_blocks:
            for r in b.resnets:
                x = r(x)
            if hasattr(b, "upsamplers"):
                x = b.upsamplers[0](x)
        x = group_norm(self.conv_norm_out, x, silu=True)
        return conv_nhwc(self.conv_out, x).permute(0, 3, 1, 2).contiguous()


class AutoencoderKL(nn.Module):
    scaling_factor = 0.18215

    def __init__(self):
        super().__init__()
        self.encoder, self.decoder = VaeEncoder(), VaeDecoder()
        self.quant_conv = nn.Conv2d(8, 8, 1)
        self.post_quant_conv = nn.Conv2d(4, 4, 1)

    @torch.no_grad()
    def encode_mode(self, x):
        """vae.encode(x).latent_dist.mode(): the mean half of the moments."""
        h = _conv1x1_nhwc(self.quant_conv, self.encoder(x))
        return h[..., :4].permute(0, 3, 1, 2).contiguous()

    @torch.no_grad()
    def decode(self, z):
        return self.decoder(_conv1x1_nhwc(self.post_quant_conv, z.permute(0, 2, 3, 1).contiguous()))


# ----------------------------------------------------------------------------------- pipeline
DEFAULT_CAMERA_EMBEDDING = torch.tensor(      # pipeline_mvdiffusion_image.py:136-148
    [[0.0, 0.0, 0.0, 1.0, 0.0], [0.0, -0.2362, 0.8125, 1.0, 0.0], [0.0, -0.1686, 1.6934, 1.0, 0.0],
     [0.0, 0.5220, 3.1406, 1.0, 0.0], [0.0, 0.6904, 4.8359, 1.0, 0.0], [0.0, 0.3733, 5.5859, 1.0, 0.0],
     [0.0, 0.0, 0.0, 0.0, 1.0], [0.0, -0.2362, 0.8125, 0.0, 1.0], [0.0, -0.1686, 1.6934, 0.0, 1.0],
     [0.0, 0.5220, 3.1406, 0.0, 1.0], [0.0, 0.6904, 4.8359, 0.0, 1.0], [0.0, 0.3733, 5.5859, 0.0, 1.0]],
    dtype=torch.float16)


class MVDiffusionImagePipeline:
    def __init__(self, unet, vae, image_encoder, scheduler=None, num_views=6):
        self.unet, self.vae, self.image_encoder = unet, vae, image_encoder
        self.scheduler = scheduler or DDIMScheduler()
        self.num_views = num_views
        self.vae_scale_factor = 8
        self.use_graph = False        # True: replay the UNet step from a captured HIP graph (see _unet_step)
        self._graph = None

    @property
    def device(self):
        return next(self.unet.parameters()).device

    @torch.no_grad()
    def _encode_image(self, images):
        """pipeline_mvdiffusion_image.py:150-182.  images (B,3,H,W) float in [0,1]; mv.py:70 hands
        them over in the weight dtype (f16).  Every row goes through the reference's 8-bit PIL
        detour (`to_pil_image`: mul(255) in the tensor's dtype, truncated — pipeline :358) before
        the CLIP preprocessing (Pillow's antialiased bicubic to 224, /255, normalise) and the VAE
        encode (k/255 -> f16 -> *2-1): mv/preprocess.py, integer arithmetic pinned to Pillow.
        The reference feeds 12 copies of one image through CLIP and the VAE (mv.py:70);
        identical rows are encoded once and broadcast."""
        dt = torch.float16
        first = images[:1]
        same = bool((images == first).all())
        src = (first if same else images).to(dt)
        u8 = PP.to_pil_u8(src)                                        # (b,H,W,3) uint8
        pix = PP.clip_pixel_values(u8).to(dt)
        emb = self.image_encoder(pixel_values=pix).image_embeds
        emb = emb.unsqueeze(1)
        lat = self.vae.encode_mode(PP.vae_input(u8, dt)) * self.vae.scaling_factor
        if same:
            emb = emb.expand(images.shape[0], -1, -1)
            lat = lat.expand(images.shape[0], -1, -1, -1)
        return emb.contiguous(), lat.contiguous()

    def _unet_step(self, model_in, t, image_embeddings, cam):
        """One UNet evaluation.  The denoising loop calls the UNet 75 times on identical shapes
        and ~600 launches each.  With `use_graph` set the forward is captured once into a HIP graph
        (static input/output buffers) and replayed: bit-identical output, but measured no faster
        on MI355X (the step is GPU-bound: 14.8 ms either way) and the capture costs ~2.7 s, so
        eager is the default."""
        if not self.use_graph:
            return self.unet(model_in, t, image_embeddings, cam)
        g = self._graph
        key = (tuple(model_in.shape), tuple(image_embeddings.shape), tuple(cam.shape))
        if g is None or g["key"] != key:
            g = {"key": key, "x": torch.empty_like(model_in), "t": torch.zeros(1, device=model_in.device,
                                                                           dtype=t.dtype),
                 "emb": torch.empty_like(image_embeddings), "cam": torch.empty_like(cam),
                 "graph": None, "out": None}
            self._graph = g
        g["x"].copy_(model_in)
        g["t"].copy_(t.reshape(1))
        g["emb"].copy_(image_embeddings)
        g["cam"].copy_(cam)
        if g["graph"] is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):                       # warm-up outside the capture
                    self.unet(g["x"], g["t"], g["emb"], g["cam"])
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                g["out"] = self.unet(g["x"], g["t"], g["emb"], g["cam"])
            g["graph"] = graph
        g["graph"].replay()
        return g["out"]

    def prepare_camera_embedding(self, camera_embedding):
        ce = camera_embedding.to(dtype=torch.float16, device=self.device)
        return torch.cat([torch.sin(ce), torch.cos(ce)], dim=-1)

    @torch.no_grad()
    def __call__(self, image, camera_embedding=None, height=256, width=256,
                 num_inference_steps=75, guidance_scale=1.0, num_images_per_prompt=1, eta=1.0,
                 generator=None, latents=None, output_type="pt", step_noise=None, callback=None):
        assert guidance_scale == 1.0 and num_images_per_prompt == 1, \
            "mv.py runs without classifier-free guidance (mv.py:81-83)"
        dev, dt = self.device, torch.float16
        image = image.to(dev)
        B = image.shape[0]
        assert B >= self.num_views and B % self.num_views == 0
        image_embeddings, image_latents = self._encode_image(image)
        if camera_embedding is None:
            camera_embedding = DEFAULT_CAMERA_EMBEDDING.repeat(B // 12, 1)
        cam = self.prepare_camera_embedding(camera_embedding)
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        shape = (B, self.unet.config["out_channels"], height // 8, width // 8)
        if latents is None:
            latents = torch.randn(shape, generator=generator, device=dev, dtype=dt)
        latents = latents.to(dev, dt) * self.scheduler.init_noise_sigma
        for i, t in enumerate(self.scheduler.timesteps):
            model_in = torch.cat([latents, image_latents], dim=1)
            noise_pred = self._unet_step(model_in, t, image_embeddings, cam)
            vn = None if step_noise is None else step_noise[i].to(dev, dt)
            # the host copy of the schedule: int(device tensor) was one blocking read-back per step
            t_host = self.scheduler.timesteps_host[i] if hasattr(self.scheduler, "timesteps_host") else t
            latents = self.scheduler.step(noise_pred, t_host, latents, eta=eta, generator=generator,
                                          variance_noise=vn)
            if callback is not None:
                callback(i, t, latents)
        if output_type == "latent":
            return latents
        img = self.vae.decode(latents / self.vae.scaling_factor)
        return (img / 2 + 0.5).clamp(0, 1)          # VaeImageProcessor.postprocess('pt')


def build_random_pipeline(device="cuda", seed=0, with_clip=True):
    """Wonder3D-joint architecture with random-init weights (no checkpoint is reachable)."""
    from .unet import UNetMV2DConditionModel
    torch.manual_seed(seed)
    unet = UNetMV2DConditionModel().half().to(device).eval()
    vae = AutoencoderKL().half().to(device).eval()
    if with_clip:
        from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
        cfg = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                               num_attention_heads=16, image_size=224, patch_size=14,
                               projection_dim=768)
        enc = CLIPVisionModelWithProjection(cfg).half().to(device).eval()
    else:
        enc = None
    return MVDiffusionImagePipeline(unet, vae, enc)
