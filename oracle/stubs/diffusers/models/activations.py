import torch.nn as nn


def get_activation(act_fn):
    table = {"swish": nn.SiLU, "silu": nn.SiLU, "mish": nn.Mish, "gelu": nn.GELU, "relu": nn.ReLU}
    if act_fn not in table:
        raise ValueError(f"Unsupported activation function: {act_fn}")
    return table[act_fn]()
