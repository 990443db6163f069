"""xformers.ops.memory_efficient_attention, documented semantics, plain torch (any dtype / CPU).
Call sites in the reference: mvdiffusion/models/transformer_mv2d.py:802,890."""
import torch


def memory_efficient_attention(query, key, value, attn_bias=None, p=0.0, scale=None, op=None):
    """(B*H, M, d) x (B*H, N, d) -> (B*H, M, d): softmax(q k^T * scale + bias) v, scale = d^-0.5."""
    assert query.dim() == 3 and key.dim() == 3 and value.dim() == 3 and p == 0.0
    scale = query.shape[-1] ** -0.5 if scale is None else scale
    out = []
    step = max(1, int(2 ** 26 // max(query.shape[1] * key.shape[1], 1)))
    for b0 in range(0, query.shape[0], step):
        sl = slice(b0, b0 + step)
        s = torch.bmm(query[sl], key[sl].transpose(1, 2)) * scale
        if attn_bias is not None:
            s = s + attn_bias[sl]
        out.append(torch.bmm(torch.softmax(s, dim=-1), value[sl]))
    return torch.cat(out, 0)
