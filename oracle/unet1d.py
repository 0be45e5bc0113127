"""Oracle restatement of the vendored diffusion-policy `ConditionalUnet1D` + a DDIM sampling loop.  TEST INFRASTRUCTURE ONLY.

Reference (in-tree, vendored): internnav/model/encoder/diffusion_policy/model/diffusion/conditional_unet1d.py:14-241
(ConditionalResidualBlock1D :14-66, ConditionalUnet1D :69-241), conv1d_components.py:7-40 (Downsample1d, Upsample1d, Conv1dBlock),
positional_embedding.py:5-17 (SinusoidalPosEmb); sampling loop as diffusion_policy/policy/diffusion_unet_lowdim_policy.py
(conditional_sample: scheduler.set_timesteps, model(trajectory, t, global_cond=...), scheduler.step). No InternNav policy instantiates
this network; north_star names it as the "third_party/diffusion-policy UNet". PINNED against the reference module
executed in the build container (oracle/make_golden.py: gold_unet1d); the DDIM scheduler is diffusers' ("parity unpinned").
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .schedulers import DDIMScheduler


def mish(x):
    return x * torch.tanh(F.softplus(x))


def conv_block(x, sd, p, n_groups=8):
    """Conv1dBlock: Conv1d(k, padding k//2) -> GroupNorm(n_groups) -> Mish on [B, C, T]."""
    w = sd[p + ".block.0.weight"]
    x = F.conv1d(x, w, sd[p + ".block.0.bias"], padding=w.shape[-1] // 2)
    x = F.group_norm(x, n_groups, sd[p + ".block.1.weight"], sd[p + ".block.1.bias"], eps=1e-5)
    return mish(x)


def res_block(x, cond, sd, p, n_groups=8):
    """ConditionalResidualBlock1D with cond_predict_scale=True (FiLM scale and bias from Mish -> Linear of the global feature)."""
    out = conv_block(x, sd, p + ".blocks.0", n_groups)
    emb = F.linear(mish(cond), sd[p + ".cond_encoder.1.weight"], sd[p + ".cond_encoder.1.bias"])
    C = out.shape[1]
    scale, bias = emb[:, :C, None], emb[:, C:, None]
    out = scale * out + bias
    out = conv_block(out, sd, p + ".blocks.1", n_groups)
    if p + ".residual_conv.weight" in sd:
        x = F.conv1d(x, sd[p + ".residual_conv.weight"], sd[p + ".residual_conv.bias"])
    return out + x


def time_embedding(t, sd, dsed=256):
    half = dsed // 2
    e = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    e = t.float()[:, None] * e[None]
    e = torch.cat([e.sin(), e.cos()], dim=-1)
    return F.linear(mish(F.linear(e, sd["diffusion_step_encoder.1.weight"], sd["diffusion_step_encoder.1.bias"])),
                    sd["diffusion_step_encoder.3.weight"], sd["diffusion_step_encoder.3.bias"])


def unet_forward(sd, sample, timestep, global_cond, n_levels=3):
    """ConditionalUnet1D.forward (:189-241) without local conditioning: sample [B, T, D] -> [B, T, D]."""
    x = sample.transpose(1, 2)
    t = torch.as_tensor(timestep).reshape(-1).expand(x.shape[0])
    g = torch.cat([time_embedding(t, sd), global_cond], dim=-1)
    h = []
    for i in range(n_levels):
        x = res_block(x, g, sd, f"down_modules.{i}.0")
        x = res_block(x, g, sd, f"down_modules.{i}.1")
        h.append(x)
        if i < n_levels - 1:
            x = F.conv1d(x, sd[f"down_modules.{i}.2.conv.weight"], sd[f"down_modules.{i}.2.conv.bias"], stride=2, padding=1)
    for i in range(2):
        x = res_block(x, g, sd, f"mid_modules.{i}")
    for i in range(n_levels - 1):
        x = torch.cat([x, h.pop()], dim=1)
        x = res_block(x, g, sd, f"up_modules.{i}.0")
        x = res_block(x, g, sd, f"up_modules.{i}.1")
        x = F.conv_transpose1d(x, sd[f"up_modules.{i}.2.conv.weight"], sd[f"up_modules.{i}.2.conv.bias"], stride=2, padding=1)   # (:171: both levels upsample)
    x = conv_block(x, sd, "final_conv.0")
    x = F.conv1d(x, sd["final_conv.1.weight"], sd["final_conv.1.bias"])
    return x.transpose(1, 2)


def ddim_sample(sd, global_cond, x_init, num_train_timesteps=100, num_inference_steps=10, use_clipped_model_output=False):
    """global_cond [B, G] (one per env), x_init [B, S, T, D] -> [B, S, T, D]: every env's S samples share its condition.
    The loop of diffusion_unet_lowdim_policy.py:77-91 (conditional_sample); use_clipped_model_output is one of its **kwargs (default: absent)."""
    B, S, T, D = x_init.shape
    sch = DDIMScheduler(num_train_timesteps=num_train_timesteps)
    sch.set_timesteps(num_inference_steps)
    x = x_init.reshape(B * S, T, D).float()
    g = global_cond.float().repeat_interleave(S, dim=0)
    for t in sch.timesteps.tolist():
        eps = unet_forward(sd, x, t, g)
        x = sch.step(eps, t, x, use_clipped_model_output=use_clipped_model_output).prev_sample
    return x.reshape(B, S, T, D)
