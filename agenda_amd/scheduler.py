"""Host-side DDIM scheduler (timesteps + alpha products); the per-step arithmetic runs on the
GPU (`agd_denoise` / `agd_cfg_ddim_step`).  Mirrors the diffusers `DDIMScheduler` config SD ships
with [upstream-knowledge, SURVEY.md §8a row S1]: scaled-linear betas, "leading" spacing,
steps_offset=1, clip_sample=False, set_alpha_to_one=False, eta=0."""
from __future__ import annotations

import numpy as np


class DDIMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 set_alpha_to_one=False, prediction_type="epsilon"):
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.prediction_type = prediction_type
        # the same torch float32 ops diffusers uses (scaled_linear betas -> cumprod), so the table is bit-identical
        import torch
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).numpy()
        self.final_alpha_cumprod = np.float32(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.num_inference_steps = None

    @classmethod
    def from_config(cls, sc):
        return cls(sc.num_train_timesteps, sc.beta_start, sc.beta_end, sc.steps_offset, sc.set_alpha_to_one,
                   sc.prediction_type)

    def set_timesteps(self, num_inference_steps: int):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps exceeds num_train_timesteps")
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.num_inference_steps = num_inference_steps
        self.timesteps = ts
        return ts

    def step_coeffs(self):
        """Per-step (alpha_cumprod[t], alpha_cumprod[prev_t]) arrays for the current timesteps."""
        ratio = self.num_train_timesteps // self.num_inference_steps
        a_t = np.array([self.alphas_cumprod[t] for t in self.timesteps], dtype=np.float32)
        a_p = np.array([self.alphas_cumprod[t - ratio] if t - ratio >= 0 else self.final_alpha_cumprod
                        for t in self.timesteps], dtype=np.float32)
        return a_t, a_p
