"""Oracle restatement of the two diffusers==0.33.1 schedulers the reference samples with.  TEST INFRASTRUCTURE ONLY.

diffusers is an un-vendored third-party dependency (requirements/internvla_n1.txt:3) that is NOT installed in this
image, so these classes restate its published algorithm ("parity unpinned"). Call sites in the reference:
  DDPMScheduler(num_train_timesteps=10|20, beta_schedule='squaredcos_cap_v2', clip_sample=True,
                prediction_type='epsilon')           navdp_policy.py:119-121, 312-315; internvla_n1/navdp.py:74-76, 247-250
  FlowMatchEulerDiscreteScheduler()                  internvla_n1.py:360, 396-397, 431

Stochasticity: DDPMScheduler.step adds sigma_t * randn for t > 0. For parity the noise is an explicit argument here
(`noise=`); both the oracle and the HIP path consume the same host-generated tensor (SURVEY.md 7 "hard parts").
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch


class DDPMScheduler:
    """DDPM (Ho et al. 2020) ancestral sampler, diffusers semantics: squaredcos_cap_v2 betas, epsilon prediction,
    clip_sample to [-1, 1], variance_type 'fixed_small', timestep_spacing 'leading'."""

    def __init__(self, num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2", clip_sample=True,
                 prediction_type="epsilon", clip_sample_range=1.0, **_):
        assert beta_schedule == "squaredcos_cap_v2" and prediction_type == "epsilon"
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, clip_sample=clip_sample,
                                      clip_sample_range=clip_sample_range, prediction_type=prediction_type)

        def alpha_bar(t):
            return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2

        betas = []
        for i in range(num_train_timesteps):
            t1, t2 = i / num_train_timesteps, (i + 1) / num_train_timesteps
            betas.append(min(1 - alpha_bar(t2) / alpha_bar(t1), 0.999))
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    def set_timesteps(self, num_inference_steps):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def coefficients(self, t: int):
        """(1/sqrt(abar_t), sqrt(1-abar_t), c_x0, c_xt, sigma_t) of one reverse step - the five scalars a fused kernel needs."""
        n_inf = self.num_inference_steps or self.config.num_train_timesteps
        prev_t = t - self.config.num_train_timesteps // n_inf
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        c_x0 = a_prev.sqrt() * cur_b / b_t
        c_xt = cur_a.sqrt() * b_prev / b_t
        var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20)
        sigma = var.sqrt() if t > 0 else torch.tensor(0.0)
        return (float(1.0 / a_t.sqrt()), float(b_t.sqrt()), float(c_x0), float(c_xt), float(sigma))

    def step(self, model_output, timestep, sample, noise=None, generator=None):
        t = int(timestep)
        inv_sqrt_a, sqrt_b, c_x0, c_xt, sigma = self.coefficients(t)
        x0 = (sample - sqrt_b * model_output) * inv_sqrt_a
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        prev = c_x0 * x0 + c_xt * sample
        if t > 0:
            if noise is None:
                noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            prev = prev + sigma * noise
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)

    def add_noise(self, original, noise, timesteps):
        a = self.alphas_cumprod[timesteps].to(original.dtype)
        sa, sb = a.sqrt(), (1 - a).sqrt()
        while sa.dim() < original.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original + sb * noise


class FlowMatchEulerDiscreteScheduler:
    """Rectified-flow Euler sampler, diffusers semantics with the default shift=1.0: x <- x + (sigma_next - sigma) * v."""

    def __init__(self, num_train_timesteps=1000, shift=1.0, **_):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift)
        # state after construction (what the SFT loss indexes, internvla_n1.py:264-268): timesteps N..1, sigmas = t / N (shifted)
        t = torch.from_numpy(np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy())
        sig = t / num_train_timesteps
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = sig * num_train_timesteps
        self.sigmas = sig
        self._step_index = None

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None):
        sig = np.asarray(sigmas, dtype=np.float32)
        s = self.config.shift
        sig = s * sig / (1 + (s - 1) * sig)
        sig_t = torch.from_numpy(sig).to(torch.float32)
        self.timesteps = sig_t * self.config.num_train_timesteps
        self.sigmas = torch.cat([sig_t, torch.zeros(1)])
        self._step_index = None

    def step(self, model_output, timestep, sample):
        if self._step_index is None:
            self._step_index = int((self.timesteps == timestep).nonzero()[0].item())
        x = sample.to(torch.float32)
        sg, sn = self.sigmas[self._step_index], self.sigmas[self._step_index + 1]
        prev = x + (sn - sg) * model_output
        self._step_index += 1
        return SimpleNamespace(prev_sample=prev.to(model_output.dtype))


class DDIMScheduler(DDPMScheduler):
    """DDIM (Song et al. 2021), diffusers semantics as configured by the vendored diffusion_policy
    (config/train_diffusion_unet_ddim_lowdim_workspace.yaml:38-49): squaredcos_cap_v2 betas, epsilon prediction, clip_sample,
    set_alpha_to_one=True, steps_offset=0, timestep_spacing 'leading', eta = 0 (deterministic: no noise term).  "parity unpinned".

    `step(..., use_clipped_model_output=False)` is diffusers' signature and default: the vendored policy calls
    `self.noise_scheduler.step(model_output, t, trajectory, generator=generator, **kwargs)` with empty kwargs
    (diffusion_policy/policy/diffusion_unet_lowdim_policy.py:87-91) and the config sets no such flag, so the direction term keeps
    the NETWORK's epsilon even when x0 was clipped; only `True` re-derives epsilon from the clipped x0 (DDIMScheduler.step, "5./6.")."""

    def coefficients(self, t: int):
        """(1/sqrt(abar_t), sqrt(1-abar_t), sqrt(abar_prev), sqrt(1-abar_prev)): x0 = clip((x - sqrt(1-abar_t) eps) / sqrt(abar_t));
        x_prev = sqrt(abar_prev) x0 + sqrt(1-abar_prev) eps, with eps the model output (default) or, under use_clipped_model_output,
        eps' = (x - sqrt(abar_t) x0) / sqrt(1-abar_t) re-derived from the clipped x0."""
        n_inf = self.num_inference_steps or self.config.num_train_timesteps
        prev_t = t - self.config.num_train_timesteps // n_inf
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one          # set_alpha_to_one
        return float(1.0 / a_t.sqrt()), float((1 - a_t).sqrt()), float(a_prev.sqrt()), float((1 - a_prev).sqrt())

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False, **_):
        assert eta == 0.0
        inv_sqrt_a, sqrt_b, sqrt_ap, sqrt_bp = self.coefficients(int(timestep))
        x0 = (sample - sqrt_b * model_output) * inv_sqrt_a
        eps = model_output
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        if use_clipped_model_output:
            eps = (sample - x0 / inv_sqrt_a) / sqrt_b
        return SimpleNamespace(prev_sample=sqrt_ap * x0 + sqrt_bp * eps, pred_original_sample=x0)
