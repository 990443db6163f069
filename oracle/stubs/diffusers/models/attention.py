"""FeedForward / GEGLU (diffusers 0.19.3 models/attention.py)."""
import torch.nn as nn
import torch.nn.functional as F

from .attention_processor import Attention  # noqa: F401  (re-exported, transformer_mv2d.py:24)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, hidden_states):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu",
                 final_dropout=False):
        super().__init__()
        inner_dim = int(dim * mult)
        dim_out = dim_out if dim_out is not None else dim
        assert activation_fn == "geglu", "diffusers stub: only GEGLU is on this path"
        self.net = nn.ModuleList([GEGLU(dim, inner_dim), nn.Dropout(dropout),
                                  nn.Linear(inner_dim, dim_out)])
        if final_dropout:
            self.net.append(nn.Dropout(dropout))

    def forward(self, hidden_states):
        for module in self.net:
            hidden_states = module(hidden_states)
        return hidden_states


class AdaLayerNorm(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("diffusers stub: AdaLayerNorm is not on the Wonder3D joint path")


class AdaLayerNormZero(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("diffusers stub: AdaLayerNormZero is not on this path")
