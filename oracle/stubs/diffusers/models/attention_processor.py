"""Attention + its default processors (diffusers 0.19.3 models/attention_processor.py): bias-free
to_q/k/v, biased to_out[0], scale = dim_head^-0.5, (batch, heads) <-> batch*heads reshapes."""
from typing import Union

import torch
import torch.nn as nn


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0,
                 bias=False, upcast_attention=False, upcast_softmax=False, cross_attention_norm=None,
                 cross_attention_norm_num_groups=32, added_kv_proj_dim=None, norm_num_groups=None,
                 spatial_norm_dim=None, out_bias=True, scale_qk=True, only_cross_attention=False,
                 eps=1e-5, rescale_output_factor=1.0, residual_connection=False,
                 _from_deprecated_attn_block=False, processor=None):
        super().__init__()
        inner_dim = dim_head * heads
        cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention = upcast_attention
        self.upcast_softmax = upcast_softmax
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.scale_qk = scale_qk
        self.scale = dim_head ** -0.5 if scale_qk else 1.0
        self.heads = heads
        self.sliceable_head_dim = heads
        self.added_kv_proj_dim = added_kv_proj_dim
        self.only_cross_attention = only_cross_attention
        assert norm_num_groups is None and spatial_norm_dim is None and cross_attention_norm is None \
            and added_kv_proj_dim is None, "diffusers stub: not on the Wonder3D joint path"
        self.group_norm = None
        self.spatial_norm = None
        self.norm_cross = None
        self.to_q = nn.Linear(query_dim, inner_dim, bias=bias)
        self.to_k = nn.Linear(cross_attention_dim, inner_dim, bias=bias)
        self.to_v = nn.Linear(cross_attention_dim, inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner_dim, query_dim, bias=out_bias),
                                     nn.Dropout(dropout)])
        self.set_processor(processor if processor is not None else AttnProcessor())

    def set_use_memory_efficient_attention_xformers(self, use, attention_op=None):
        self.set_processor(XFormersAttnProcessor(attention_op) if use else AttnProcessor())

    def set_processor(self, processor):
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None,
                **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)

    def batch_to_head_dim(self, tensor):
        head_size = self.heads
        batch_size, seq_len, dim = tensor.shape
        tensor = tensor.reshape(batch_size // head_size, head_size, seq_len, dim)
        return tensor.permute(0, 2, 1, 3).reshape(batch_size // head_size, seq_len, dim * head_size)

    def head_to_batch_dim(self, tensor, out_dim=3):
        head_size = self.heads
        batch_size, seq_len, dim = tensor.shape
        tensor = tensor.reshape(batch_size, seq_len, head_size, dim // head_size)
        tensor = tensor.permute(0, 2, 1, 3)
        if out_dim == 3:
            tensor = tensor.reshape(batch_size * head_size, seq_len, dim // head_size)
        return tensor

    def get_attention_scores(self, query, key, attention_mask=None):
        dtype = query.dtype
        if self.upcast_attention:
            query, key = query.float(), key.float()
        scores = torch.bmm(query, key.transpose(-1, -2)) * self.scale
        if attention_mask is not None:
            scores = scores + attention_mask
        if self.upcast_softmax:
            scores = scores.float()
        return scores.softmax(dim=-1).to(dtype)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size=None, out_dim=3):
        if attention_mask is None:
            return None
        raise NotImplementedError("diffusers stub: attention masks are not on this path "
                                  "(transformer_mv2d.py:541 asserts None)")


class AttnProcessor:
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        residual = hidden_states
        assert hidden_states.ndim == 3
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        query = attn.head_to_batch_dim(query)
        key = attn.head_to_batch_dim(key)
        value = attn.head_to_batch_dim(value)
        probs = attn.get_attention_scores(query, key, attention_mask)
        hidden_states = attn.batch_to_head_dim(torch.bmm(probs, value))
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        if attn.residual_connection:
            hidden_states = hidden_states + residual
        return hidden_states / attn.rescale_output_factor


class XFormersAttnProcessor:
    def __init__(self, attention_op=None):
        self.attention_op = attention_op

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        import xformers.ops
        residual = hidden_states
        assert hidden_states.ndim == 3 and attention_mask is None
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        query = attn.head_to_batch_dim(query).contiguous()
        key = attn.head_to_batch_dim(key).contiguous()
        value = attn.head_to_batch_dim(value).contiguous()
        hidden_states = xformers.ops.memory_efficient_attention(query, key, value, attn_bias=None,
                                                                scale=attn.scale)
        hidden_states = attn.batch_to_head_dim(hidden_states.to(query.dtype))
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        if attn.residual_connection:
            hidden_states = hidden_states + residual
        return hidden_states / attn.rescale_output_factor


AttentionProcessor = Union[AttnProcessor, XFormersAttnProcessor]
