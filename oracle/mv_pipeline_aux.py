"""Small fixed parts shared by tests/golden/make_mv_pipeline_golden.py and the tests that read its
fixture.  TEST INFRASTRUCTURE ONLY.  LinearVAE / LinearClip stand in for the VAE and the CLIP vision
tower inside the pipeline-glue fixture (both have the interface the REFERENCE pipeline uses AND the
one drawingspinup_amd.mv.pipeline uses); inputs and "random" draws are pure functions of a tag."""
import torch

from . import mv_weights


class LinearVAE(torch.nn.Module):
    """fixed linear 'autoencoder' with the interface the pipeline uses (encode().latent_dist.mode(),
    decode(z, return_dict=False)[0], config.scaling_factor / block_out_channels, dtype)."""

    def __init__(self):
        super().__init__()
        self.enc = torch.nn.Conv2d(3, 8, 1)
        self.dec = torch.nn.Conv2d(4, 3 * 64, 1)
        self.config = type("C", (), {"scaling_factor": 0.18215, "block_out_channels": (1, 2, 3, 4)})()

    @property
    def dtype(self):
        return self.enc.weight.dtype

    def encode(self, x):
        m = self.enc(torch.nn.functional.avg_pool2d(x, 8))
        dist = type("D", (), {"mode": staticmethod(lambda: m[:, :4])})()
        return type("E", (), {"latent_dist": dist})()

    def decode(self, z, return_dict=True):
        out = torch.nn.functional.pixel_shuffle(self.dec(z), 8)
        return (out,) if return_dict is False else out

    # the product pipeline's VAE interface (drawingspinup_amd/mv/pipeline.py: AutoencoderKL)
    @property
    def scaling_factor(self):
        return self.config.scaling_factor

    def encode_mode(self, x):
        return self.encode(x).latent_dist.mode()


class LinearClip(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.proj = torch.nn.Linear(192, 768)

    def forward(self, pixel_values):
        f = torch.nn.functional.avg_pool2d(pixel_values, 28).flatten(1)
        return type("O", (), {"image_embeds": self.proj(f)})()


def aux_state(module, prefix):
    names = [(prefix + k, tuple(v.shape)) for k, v in module.state_dict().items()]
    sd = mv_weights.synth_state_dict(names)
    module.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    return module


def input_image():
    """(3,256,256) f16 in [0,1], arbitrary (not k/255) values: the to_pil_image truncation acts."""
    u = mv_weights.det_tensor("in.image.coarse", (3, 32, 32), 0.5, 0.5)
    img = torch.nn.functional.interpolate(u[None], size=(256, 256), mode="bilinear", align_corners=False)[0]
    img = (img + mv_weights.det_tensor("in.image.fine", (3, 256, 256), 0.04)).clamp(0, 1)
    return img.half()


def camera_embeddings():
    """mv.py:72-75: [elev_cond, d_elev, d_azim] per view (x2 domains) | task one-hot."""
    cam = mv_weights.det_tensor("in.camera", (6, 3), 1.5)
    cam = torch.cat([cam, cam], 0)
    task = torch.cat([torch.tensor([[1.0, 0.0]]).expand(6, 2), torch.tensor([[0.0, 1.0]]).expand(6, 2)], 0)
    return torch.cat([cam, task.double()], -1).half()


def det_noise(tag, shape):
    return (mv_weights.det_tensor(tag, shape, 3.0 ** 0.5)).half().double()
