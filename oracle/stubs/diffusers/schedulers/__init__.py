"""DDIMScheduler (diffusers 0.19.3 schedulers/scheduling_ddim.py) with the Stable-Diffusion-1.x
configuration: scaled_linear betas, 'leading' timestep spacing, steps_offset 1, epsilon prediction,
clip_sample False, set_alpha_to_one False.  Arithmetic in the sample's dtype (float64 here)."""
import enum
from dataclasses import dataclass

import numpy as np
import torch

from ..utils import BaseOutput, randn_tensor


class KarrasDiffusionSchedulers(enum.Enum):
    DDIMScheduler = 1


@dataclass
class DDIMSchedulerOutput(BaseOutput):
    prev_sample: torch.FloatTensor
    pred_original_sample: torch.FloatTensor = None


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False,
                 steps_offset=1, prediction_type="epsilon", timestep_spacing="leading"):
        assert beta_schedule == "scaled_linear" and prediction_type == "epsilon" and not clip_sample \
            and timestep_spacing == "leading"
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                    dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_train_timesteps, self.steps_offset = num_train_timesteps, steps_offset
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        step_ratio = self.num_train_timesteps // self.num_inference_steps
        timesteps = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        timesteps += self.steps_offset
        self.timesteps = torch.from_numpy(timesteps).to(device)

    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False,
             generator=None, variance_noise=None, return_dict=True):
        timestep = int(timestep)
        prev_timestep = timestep - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep].to(sample.dtype)
        a_prev = (self.alphas_cumprod[prev_timestep] if prev_timestep >= 0
                  else self.final_alpha_cumprod).to(sample.dtype)
        beta_prod_t = 1 - a_t
        pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / a_t ** 0.5
        pred_epsilon = model_output
        variance = self._get_variance(timestep, prev_timestep).to(sample.dtype)
        std_dev_t = eta * variance ** 0.5
        pred_sample_direction = (1 - a_prev - std_dev_t ** 2) ** 0.5 * pred_epsilon
        prev_sample = a_prev ** 0.5 * pred_original_sample + pred_sample_direction
        if eta > 0:
            if variance_noise is None:
                variance_noise = randn_tensor(model_output.shape, generator=generator,
                                              device=model_output.device, dtype=model_output.dtype)
            prev_sample = prev_sample + std_dev_t * variance_noise
        return DDIMSchedulerOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample)
