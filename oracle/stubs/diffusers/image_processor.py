"""VaeImageProcessor.postprocess for output_type 'pt' / 'latent' (diffusers 0.19.3
image_processor.py): per-image denormalize = (x / 2 + 0.5).clamp(0, 1)."""
import torch


class VaeImageProcessor:
    def __init__(self, do_resize=True, vae_scale_factor=8, resample="lanczos", do_normalize=True,
                 do_convert_rgb=False):
        self.vae_scale_factor = vae_scale_factor

    @staticmethod
    def denormalize(images):
        return (images / 2 + 0.5).clamp(0, 1)

    def postprocess(self, image, output_type="pil", do_denormalize=None):
        assert torch.is_tensor(image)
        if output_type == "latent":
            return image
        if do_denormalize is None:
            do_denormalize = [True] * image.shape[0]
        image = torch.stack([self.denormalize(image[i]) if do_denormalize[i] else image[i]
                             for i in range(image.shape[0])])
        if output_type == "pt":
            return image
        raise NotImplementedError("diffusers stub: output_type 'pt' / 'latent' only (mv.py:83)")
