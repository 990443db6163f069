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
PARITY UNPINNED: the reference ships no tests/golden vectors for this path and its
third-party pieces cannot be executed here; KATs are hand-derived (tests/test_oracle_mv.py).
Everything is float64 on the CPU.
"""
import math

import torch
import torch.nn.functional as F
from einops import rearrange, repeat


def memory_efficient_attention(q, k, v):
    """xformers semantics on (B*H, M, d): softmax(q k^T * d^-0.5) v."""
    s = torch.einsum("bmd,bnd->bmn", q, k) * (q.shape[-1] ** -0.5)
    return torch.einsum("bmn,bnd->bmd", torch.softmax(s, dim=-1), v)


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
