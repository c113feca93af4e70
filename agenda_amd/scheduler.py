"""Host-side schedulers (timesteps + alpha products); the per-step arithmetic runs on the GPU
(`agd_denoise` / `agd_cfg_ddim_step` for DDIM, `agd_denoise_plms` for PNDM).

DDIM is BASELINE.json's metric (50 DDIM steps).  PNDM (skip_prk_steps = PLMS) is what the reference itself runs:
`pipeline(prompt, num_inference_steps=20)` at data_generation.py:59 never constructs a scheduler, so the fine-tuned
CompVis SD-1.4 checkpoint's `scheduler/scheduler_config.json` (`PNDMScheduler`) decides [upstream-knowledge].
Both mirror the config SD ships with [upstream-knowledge, SURVEY.md §8a row S1]: scaled-linear betas, "leading" spacing,
steps_offset=1, clip_sample=False, set_alpha_to_one=False; DDIM eta=0."""
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


class PNDMScheduler:
    """diffusers `PNDMScheduler` as SD-1.x checkpoints configure it [upstream-knowledge: diffusers 0.21.2]: scaled-linear betas,
    `skip_prk_steps=True` (pure PLMS), "leading" spacing, `steps_offset=1`, `set_alpha_to_one=False`, epsilon prediction.
    `num_inference_steps` = n means n + 1 model evaluations: the second timestep is evaluated twice."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1, set_alpha_to_one=False,
                 prediction_type="epsilon", skip_prk_steps=True):
        if prediction_type != "epsilon":
            raise ValueError("PNDMScheduler: only epsilon prediction is implemented (what SD-1.x checkpoints use)")
        if not skip_prk_steps:
            raise ValueError("PNDMScheduler: only skip_prk_steps=True (the SD configuration) is implemented")
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.prediction_type = prediction_type
        import torch
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).numpy()
        self.final_alpha_cumprod = np.float32(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.num_inference_steps = None

    @classmethod
    def from_config(cls, sc):
        return cls(sc.num_train_timesteps, sc.beta_start, sc.beta_end, sc.steps_offset, sc.set_alpha_to_one, sc.prediction_type,
                   getattr(sc, "skip_prk_steps", True))

    def set_timesteps(self, num_inference_steps: int):
        if num_inference_steps < 2 or num_inference_steps > self.num_train_timesteps:
            raise ValueError("PNDM needs 2 <= num_inference_steps <= num_train_timesteps")
        ratio = self.num_train_timesteps // num_inference_steps
        t = (np.arange(0, num_inference_steps) * ratio).round().astype(np.int64) + self.steps_offset
        # plms_timesteps = concat(t[:-1], t[-2:-1], t[-1:])[::-1]: the second (in run order) timestep appears twice
        self.timesteps = np.concatenate([t[:-1], t[-2:-1], t[-1:]])[::-1].copy()
        self.num_inference_steps = num_inference_steps
        return self.timesteps

    def plms_program(self):
        """Per model evaluation i: (UNet timestep, sample_coeff, eps_coeff) of `_get_prev_sample`:
        prev = sample_coeff * sample + eps_coeff * model_output (model_output = the PLMS combination, applied on the device)."""
        ratio = self.num_train_timesteps // self.num_inference_steps
        a, b = [], []
        for i, t in enumerate(self.timesteps):
            t, prev = int(t), int(t) - ratio
            if i == 1:                                   # step_plms with counter == 1: redo the first step from the kept sample
                prev, t = t, t + ratio
            al_t = float(self.alphas_cumprod[t])
            al_p = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
            denom = al_t * (1 - al_p) ** 0.5 + (al_t * (1 - al_t) * al_p) ** 0.5
            a.append((al_p / al_t) ** 0.5)
            b.append(-(al_p - al_t) / denom)
        return np.asarray(self.timesteps, dtype=np.float32), np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)


SCHEDULERS = {"DDIMScheduler": DDIMScheduler, "PNDMScheduler": PNDMScheduler}
