"""Oracle, stage 3b: PNDMScheduler as configured for SD-1.5 (scheduler_config.json: beta_schedule
"scaled_linear" 0.00085 -> 0.012, 1000 train steps, skip_prk_steps=True, steps_offset=1,
set_alpha_to_one=False, prediction_type "epsilon").

PARITY UNPINNED: the class lives in diffusers==0.17.1 (requirements.txt:9), absent from the reference tree
and from this image.  Call sites it must serve: scheduler.set_timesteps / .timesteps / .init_noise_sigma /
.scale_model_input / .step at gill/custom_sd.py:607-608, :472, :631, :646.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch


class PNDMSchedulerRef:
  init_noise_sigma = 1.0
  order = 1

  def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
               steps_offset: int = 1, prediction_type: str = "epsilon"):
    self.num_train_timesteps = num_train_timesteps
    self.prediction_type = prediction_type            # "v_prediction": SD-2.1-768
    self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    self.alphas = 1.0 - self.betas
    self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
    self.final_alpha_cumprod = self.alphas_cumprod[0]   # set_alpha_to_one=False
    self.steps_offset = steps_offset
    self.timesteps: List[int] = []
    self.ets: List[torch.Tensor] = []
    self.counter = 0
    self.cur_sample = None

  def set_timesteps(self, num_inference_steps: int):
    self.num_inference_steps = num_inference_steps
    step_ratio = self.num_train_timesteps // num_inference_steps
    base = (np.arange(0, num_inference_steps) * step_ratio).round() + self.steps_offset
    # skip_prk_steps: no Runge-Kutta warm-up; the second-to-last timestep is repeated once
    plms = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
    self.timesteps = [int(t) for t in plms]
    self.ets, self.counter, self.cur_sample = [], 0, None
    return self.timesteps

  def scale_model_input(self, sample, t=None):
    return sample

  def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor) -> torch.Tensor:
    ratio = self.num_train_timesteps // self.num_inference_steps
    prev_timestep = timestep - ratio
    if self.counter != 1:
      self.ets = self.ets[-3:]
      self.ets.append(model_output)
    else:
      prev_timestep = timestep
      timestep = timestep + ratio
    if len(self.ets) == 1 and self.counter == 0:
      self.cur_sample = sample
    elif len(self.ets) == 1 and self.counter == 1:
      model_output = (model_output + self.ets[-1]) / 2
      sample = self.cur_sample
      self.cur_sample = None
    elif len(self.ets) == 2:
      model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
    elif len(self.ets) == 3:
      model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
    else:
      model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
    prev_sample = self._get_prev_sample(sample, timestep, prev_timestep, model_output)
    self.counter += 1
    return prev_sample

  def _get_prev_sample(self, sample, timestep, prev_timestep, model_output):
    a_t = self.alphas_cumprod[timestep]
    a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
    b_t, b_p = 1 - a_t, 1 - a_p
    if self.prediction_type == "v_prediction":        # diffusers PNDMScheduler._get_prev_sample
      model_output = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
    sample_coeff = (a_p / a_t) ** 0.5
    denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
    return sample_coeff * sample - (a_p - a_t) * model_output / denom
