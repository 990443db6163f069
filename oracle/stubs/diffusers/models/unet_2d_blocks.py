"""DownBlock2D / UpBlock2D (diffusers 0.19.3 models/unet_2d_blocks.py) — the two attention-free
blocks of the SD-1.x layout; every other block name the reference imports is a placeholder."""
import torch
import torch.nn as nn

from .resnet import Downsample2D, ResnetBlock2D, Upsample2D


class DownBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, dropout=0.0, num_layers=1,
                 resnet_eps=1e-6, resnet_time_scale_shift="default", resnet_act_fn="swish",
                 resnet_groups=32, resnet_pre_norm=True, output_scale_factor=1.0,
                 add_downsample=True, downsample_padding=1):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=in_channels if i == 0 else out_channels,
                          out_channels=out_channels, temb_channels=temb_channels, eps=resnet_eps,
                          groups=resnet_groups, dropout=dropout,
                          time_embedding_norm=resnet_time_scale_shift, non_linearity=resnet_act_fn,
                          output_scale_factor=output_scale_factor, pre_norm=resnet_pre_norm)
            for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([
            Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                         padding=downsample_padding, name="op")]) if add_downsample else None
        self.gradient_checkpointing = False

    def forward(self, hidden_states, temb=None):
        output_states = ()
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, temb)
            output_states = output_states + (hidden_states,)
        if self.downsamplers is not None:
            for downsampler in self.downsamplers:
                hidden_states = downsampler(hidden_states)
            output_states = output_states + (hidden_states,)
        return hidden_states, output_states


class UpBlock2D(nn.Module):
    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, dropout=0.0,
                 num_layers=1, resnet_eps=1e-6, resnet_time_scale_shift="default",
                 resnet_act_fn="swish", resnet_groups=32, resnet_pre_norm=True,
                 output_scale_factor=1.0, add_upsample=True):
        super().__init__()
        resnets = []
        for i in range(num_layers):
            res_skip_channels = in_channels if (i == num_layers - 1) else out_channels
            resnet_in_channels = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock2D(in_channels=resnet_in_channels + res_skip_channels,
                                         out_channels=out_channels, temb_channels=temb_channels,
                                         eps=resnet_eps, groups=resnet_groups, dropout=dropout,
                                         time_embedding_norm=resnet_time_scale_shift,
                                         non_linearity=resnet_act_fn,
                                         output_scale_factor=output_scale_factor,
                                         pre_norm=resnet_pre_norm))
        self.resnets = nn.ModuleList(resnets)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, use_conv=True,
                                                    out_channels=out_channels)]) if add_upsample else None
        self.gradient_checkpointing = False

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, upsample_size=None):
        for resnet in self.resnets:
            res_hidden_states = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res_hidden_states], dim=1)
            hidden_states = resnet(hidden_states, temb)
        if self.upsamplers is not None:
            for upsampler in self.upsamplers:
                hidden_states = upsampler(hidden_states, upsample_size)
        return hidden_states


def _unused(name):
    class _Unused(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"diffusers stub: {name} is not on the Wonder3D joint path")
    _Unused.__name__ = name
    return _Unused


for _n in ["ResnetDownsampleBlock2D", "AttnDownBlock2D", "CrossAttnDownBlock2D",
           "SimpleCrossAttnDownBlock2D", "SkipDownBlock2D", "AttnSkipDownBlock2D",
           "DownEncoderBlock2D", "AttnDownEncoderBlock2D", "KDownBlock2D", "KCrossAttnDownBlock2D",
           "ResnetUpsampleBlock2D", "CrossAttnUpBlock2D", "SimpleCrossAttnUpBlock2D",
           "AttnUpBlock2D", "SkipUpBlock2D", "AttnSkipUpBlock2D", "UpDecoderBlock2D",
           "AttnUpDecoderBlock2D", "KUpBlock2D", "KCrossAttnUpBlock2D", "UNetMidBlock2DCrossAttn",
           "UNetMidBlock2DSimpleCrossAttn"]:
    globals()[_n] = _unused(_n)
