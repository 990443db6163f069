"""ResnetBlock2D / Downsample2D / Upsample2D (diffusers 0.19.3 models/resnet.py), the
`time_embedding_norm="default"` branch."""
import torch.nn as nn
import torch.nn.functional as F

from .activations import get_activation
from .lora import LoRACompatibleConv


class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None,
                 name="conv"):
        super().__init__()
        assert use_conv and not use_conv_transpose and name == "conv"
        self.channels, self.out_channels = channels, out_channels or channels
        self.conv = LoRACompatibleConv(self.channels, self.out_channels, 3, padding=1)

    def forward(self, hidden_states, output_size=None):
        assert hidden_states.shape[1] == self.channels
        if output_size is None:
            hidden_states = F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        else:
            hidden_states = F.interpolate(hidden_states, size=output_size, mode="nearest")
        return self.conv(hidden_states)


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        assert use_conv
        self.channels, self.out_channels, self.padding = channels, out_channels or channels, padding
        conv = LoRACompatibleConv(self.channels, self.out_channels, 3, stride=2, padding=padding)
        if name == "conv":
            self.Conv2d_0 = conv
        self.conv = conv

    def forward(self, hidden_states):
        assert hidden_states.shape[1] == self.channels
        if self.padding == 0:
            hidden_states = F.pad(hidden_states, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(hidden_states)


class ResnetBlock2D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0,
                 temb_channels=512, groups=32, groups_out=None, pre_norm=True, eps=1e-6,
                 non_linearity="swish", skip_time_act=False, time_embedding_norm="default",
                 kernel=None, output_scale_factor=1.0, use_in_shortcut=None, up=False, down=False,
                 conv_shortcut_bias=True, conv_2d_out_channels=None):
        super().__init__()
        assert time_embedding_norm == "default" and not up and not down and kernel is None
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.output_scale_factor = output_scale_factor
        self.skip_time_act = skip_time_act
        groups_out = groups if groups_out is None else groups_out
        self.norm1 = nn.GroupNorm(num_groups=groups, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = LoRACompatibleConv(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(num_groups=groups_out, num_channels=out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        conv_2d_out_channels = conv_2d_out_channels or out_channels
        self.conv2 = LoRACompatibleConv(out_channels, conv_2d_out_channels, kernel_size=3, stride=1,
                                        padding=1)
        self.nonlinearity = get_activation(non_linearity)
        self.use_in_shortcut = in_channels != conv_2d_out_channels if use_in_shortcut is None \
            else use_in_shortcut
        self.conv_shortcut = None
        if self.use_in_shortcut:
            self.conv_shortcut = LoRACompatibleConv(in_channels, conv_2d_out_channels, kernel_size=1,
                                                    stride=1, padding=0, bias=conv_shortcut_bias)

    def forward(self, input_tensor, temb):
        hidden_states = self.nonlinearity(self.norm1(input_tensor))
        hidden_states = self.conv1(hidden_states)
        if self.time_emb_proj is not None:
            if not self.skip_time_act:
                temb = self.nonlinearity(temb)
            temb = self.time_emb_proj(temb)[:, :, None, None]
        if temb is not None:
            hidden_states = hidden_states + temb
        hidden_states = self.nonlinearity(self.norm2(hidden_states))
        hidden_states = self.conv2(self.dropout(hidden_states))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + hidden_states) / self.output_scale_factor
