"""Timesteps / TimestepEmbedding (diffusers 0.19.3 models/embeddings.py).  The sinusoid is
evaluated in float32 exactly as diffusers does ("`Timesteps` ... will always return f32 tensors",
unet_mv2d_condition.py:867-870); the caller casts to the sample dtype."""
import math

import torch
import torch.nn as nn

from .activations import get_activation


def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1,
                           scale=1, max_period=10000):
    assert len(timesteps.shape) == 1
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half_dim, dtype=torch.float32,
                                                    device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels,
                                      flip_sin_to_cos=self.flip_sin_to_cos,
                                      downscale_freq_shift=self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None,
                 cond_proj_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.cond_proj = nn.Linear(cond_proj_dim, in_channels, bias=False) if cond_proj_dim else None
        self.act = get_activation(act_fn)
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)
        self.post_act = None if post_act_fn is None else get_activation(post_act_fn)

    def forward(self, sample, condition=None):
        if condition is not None:
            sample = sample + self.cond_proj(condition)
        sample = self.linear_1(sample)
        if self.act is not None:
            sample = self.act(sample)
        sample = self.linear_2(sample)
        if self.post_act is not None:
            sample = self.post_act(sample)
        return sample


def _unused(name):
    class _Unused(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"diffusers stub: {name} is not on the Wonder3D joint path")
    _Unused.__name__ = name
    return _Unused


GaussianFourierProjection = _unused("GaussianFourierProjection")
ImageHintTimeEmbedding = _unused("ImageHintTimeEmbedding")
ImageProjection = _unused("ImageProjection")
ImageTimeEmbedding = _unused("ImageTimeEmbedding")
TextImageProjection = _unused("TextImageProjection")
TextImageTimeEmbedding = _unused("TextImageTimeEmbedding")
TextTimeEmbedding = _unused("TextTimeEmbedding")
ImagePositionalEmbeddings = _unused("ImagePositionalEmbeddings")
PatchEmbed = _unused("PatchEmbed")
