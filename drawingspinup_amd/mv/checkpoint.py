"""Loading the reference's diffusion checkpoint (2_charactor_reconstructor/mv.py:29-39:
`DiffusionPipeline.from_pretrained('flamehaze1115/wonder3d-v1.0', custom_pipeline=...)`) into the
modules of this package, from a LOCAL diffusers-layout directory (there is no hub access):

    <dir>/unet/diffusion_pytorch_model.{safetensors,bin}
    <dir>/vae/diffusion_pytorch_model.{safetensors,bin}
    <dir>/image_encoder/{model.safetensors,pytorch_model.bin} + config.json   (CLIP vision tower)

  * UNet keys go through the renames of UNetMV2DConditionModel.from_pretrained_2d
    (mvdiffusion/models/unet_mv2d_condition.py:1318-1332): attn_joint -> attn_joint_last,
    norm_joint -> norm_joint_last, attn_joint_twice -> attn_joint_mid,
    norm_joint_twice -> norm_joint_mid.
  * VAE attention keys of pre-0.15 checkpoints (query / key / value / proj_attn, as the SD-1.x VAE
    ships) are mapped to to_q / to_k / to_v / to_out.0 as diffusers'
    `_convert_deprecated_attention_blocks` does.
Loads are strict: a missing or unexpected key is an error.
"""
import os

import torch


def _load_file(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu")


def find_weights(folder, names=("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin",
                                "diffusion_pytorch_model.fp16.safetensors", "model.safetensors",
                                "pytorch_model.bin")):
    for n in names:
        p = os.path.join(folder, n)
        if os.path.isfile(p):
            return p
    raise FileNotFoundError(f"no weights file in {folder} (looked for {', '.join(names)})")


def rename_wonder3d_unet_keys(state_dict):
    """from_pretrained_2d's key renames (unet_mv2d_condition.py:1318-1332)."""
    out = {}
    for k, v in state_dict.items():
        if "attn_joint_twice." in k:
            k = k.replace("attn_joint_twice.", "attn_joint_mid.")
        elif "norm_joint_twice." in k:
            k = k.replace("norm_joint_twice.", "norm_joint_mid.")
        elif "attn_joint." in k:
            k = k.replace("attn_joint.", "attn_joint_last.")
        elif "norm_joint." in k:
            k = k.replace("norm_joint.", "norm_joint_last.")
        out[k] = v
    return out


_DEPRECATED_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def convert_deprecated_vae_attention(state_dict):
    out = {}
    for k, v in state_dict.items():
        parts = k.split(".")
        if "attentions" in parts:
            i = parts.index("attentions")
            if len(parts) > i + 2 and parts[i + 2] in _DEPRECATED_ATTN:
                parts[i + 2] = _DEPRECATED_ATTN[parts[i + 2]]
                k = ".".join(parts)
                if v.dim() == 4:                       # 1x1 conv weights of very old checkpoints
                    v = v[:, :, 0, 0]
        out[k] = v
    return out


def load_unet(unet, path):
    sd = rename_wonder3d_unet_keys(_load_file(path))
    unet.load_state_dict({k: v.to(torch.float16) for k, v in sd.items()}, strict=True)
    return unet


def load_vae(vae, path):
    sd = convert_deprecated_vae_attention(_load_file(path))
    vae.load_state_dict({k: v.to(torch.float16) for k, v in sd.items()}, strict=True)
    return vae


def load_image_encoder(folder, device):
    from transformers import CLIPVisionModelWithProjection
    return CLIPVisionModelWithProjection.from_pretrained(folder, local_files_only=True) \
        .half().to(device).eval()


def load_pipeline(folder, device):
    """The three modules of the Wonder3D checkpoint from a local directory."""
    from .pipeline import AutoencoderKL, MVDiffusionImagePipeline
    from .unet import UNetMV2DConditionModel
    unet = load_unet(UNetMV2DConditionModel(), find_weights(os.path.join(folder, "unet")))
    vae = load_vae(AutoencoderKL(), find_weights(os.path.join(folder, "vae")))
    enc = load_image_encoder(os.path.join(folder, "image_encoder"), device)
    return MVDiffusionImagePipeline(unet.half().to(device).eval(), vae.half().to(device).eval(), enc)
