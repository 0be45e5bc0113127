"""Oracle restatement of the InternVLA-N1 `nextdit_async` System-1 call (DualVLN checkpoint).  TEST INFRASTRUCTURE ONLY.

Follows
  internnav/model/basemodel/internvla_n1/internvla_n1.py:349-432      generate_traj, 'nextdit' + 'async' branch
  internnav/model/basemodel/internvla_n1/internvla_n1_arch.py:50-118  SinusoidalPositionalEncoding, MemoryEncoder, QFormer
  internnav/model/basemodel/internvla_n1/internvla_n1_arch.py:127-141 action_encoder / action_decoder / cond_projector
  internnav/model/basemodel/internvla_n1/nextdit_traj.py:121-178      LuminaNextDiTBlock.forward
  internnav/model/basemodel/internvla_n1/nextdit_traj.py:299-368      LuminaNextDiT2DModel.forward
  internnav/model/basemodel/internvla_n1/nextdit_crossattn_traj.py:86-96 NextDiTCrossAttn.forward (no rotary, all-ones mask)
and, for the un-vendored diffusers==0.33.1 blocks those files import, the published semantics restated in
oracle/diffusers_blocks.py ("parity unpinned").

State-dict prefixes = attribute names on InternVLAN1Model (a real checkpoint carries them under `model.`):
  cond_projector.{0,2}, rgb_model.*, memory_encoder.*, rgb_resampler.*, action_encoder, action_decoder, traj_dit.model.*
The reference runs one environment (traj_latents [1,4,3584]) and 32 samples per call; classifier-free guidance runs a
null-condition half that is multiplied by zero weight when guidance_scale == 1.0 (internvla_n1.py:427-428):
noise_pred = uncond + 1.0 * (cond - uncond). The oracle evaluates exactly that expression (both halves).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import dinov2
from .nn_ref import decoder_layer, encoder_layer, layer_norm, linear, rms_norm, sdpa

RESNET_MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
RESNET_STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
DIT_HEADS = 6


def sinusoidal_positional_encoding(T: int, dim: int) -> torch.Tensor:
    """SinusoidalPositionalEncoding(dim)(arange(T)) (internvla_n1_arch.py:50-73)."""
    half = dim // 2
    exponent = -torch.arange(half, dtype=torch.float32) * (torch.log(torch.tensor(10000.0)) / half)
    freqs = torch.arange(T, dtype=torch.float32).unsqueeze(-1) * exponent.exp()
    return torch.cat([torch.sin(freqs), torch.cos(freqs)], dim=-1)


def timestep_embedding(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    """diffusers Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin] of t * 10000^(-i/128)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    a = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


def memory_encoder(x, sd, p="memory_encoder."):
    """MemoryEncoder.forward (internvla_n1_arch.py:85-94): + learned pos, 3 post-LN encoder layers (6 heads, ReLU ffn 2048)."""
    x = x + sd[p + "memory_pos"][: x.shape[1]]
    for i in range(3):
        x = encoder_layer(x, sd, f"{p}encoder.layers.{i}", 6)
    return x


def qformer(mem, sd, p="rgb_resampler."):
    """QFormer.forward (internvla_n1_arch.py:110-118): 32 learned queries, 3 post-LN decoder layers (12 heads, ReLU ffn 2048)."""
    q = (sd[p + "query_tokens"] + sd[p + "query_pos"]).unsqueeze(0).expand(mem.shape[0], -1, -1)
    for i in range(3):
        q = decoder_layer(q, mem, sd, f"{p}decoder.layers.{i}", 12, norm_first=False, act="relu")
    return q


def _dit_attention(x_q, x_kv, sd, p):
    """diffusers Attention + LuminaAttnProcessor2_0, qk_norm='layer_norm_across_heads', no rotary, no mask, no out-proj."""
    B, L, C = x_q.shape
    q = layer_norm(F.linear(x_q, sd[p + ".to_q.weight"]), sd, p + ".norm_q", 1e-5)
    k = layer_norm(F.linear(x_kv, sd[p + ".to_k.weight"]), sd, p + ".norm_k", 1e-5)
    v = F.linear(x_kv, sd[p + ".to_v.weight"])
    hd = C // DIT_HEADS
    q = q.view(B, L, DIT_HEADS, hd).transpose(1, 2)
    k = k.view(B, -1, DIT_HEADS, hd).transpose(1, 2)
    v = v.view(B, -1, DIT_HEADS, hd).transpose(1, 2)
    return sdpa(q, k, v).transpose(1, 2)  # [B, L, H, hd]


def dit_block(x, enc, temb, sd, p):
    """LuminaNextDiTBlock.forward (nextdit_traj.py:121-178)."""
    emb = linear(F.silu(temb), sd, p + ".norm1.linear")
    scale_msa, gate_msa, scale_mlp, gate_mlp = emb.chunk(4, dim=1)
    nh = rms_norm(x, sd[p + ".norm1.norm.weight"], 1e-5) * (1 + scale_msa[:, None])
    sa = _dit_attention(nh, nh, sd, p + ".attn1")
    ca = _dit_attention(nh, rms_norm(enc, sd[p + ".norm1_context.weight"], 1e-5), sd, p + ".attn2")
    ca = ca * sd[p + ".gate"].tanh().view(1, 1, -1, 1)
    mixed = (sa + ca).flatten(-2)
    h = F.linear(mixed, sd[p + ".attn2.to_out.0.weight"])
    x = x + gate_msa.unsqueeze(1).tanh() * rms_norm(h, sd[p + ".norm2.weight"], 1e-5)
    y = rms_norm(x, sd[p + ".ffn_norm1.weight"], 1e-5) * (1 + scale_mlp.unsqueeze(1))
    y = F.linear(F.silu(F.linear(y, sd[p + ".feed_forward.linear_1.weight"])) * F.linear(y, sd[p + ".feed_forward.linear_3.weight"]),
                 sd[p + ".feed_forward.linear_2.weight"])
    return x + gate_mlp.unsqueeze(1).tanh() * rms_norm(y, sd[p + ".ffn_norm2.weight"], 1e-5)


def traj_dit(x, timestep, z, sd, p="traj_dit.model.", n_layers=12):
    """LuminaNextDiT2DModel.forward (nextdit_traj.py:299-368): x [N,T,384], timestep [N], z [N,L,768] -> [N,T,384]."""
    enc = linear(F.gelu(linear(z, sd, p + "caption_projection.linear_1"), approximate="tanh"), sd, p + "caption_projection.linear_2")
    te = linear(F.silu(linear(timestep_embedding(timestep), sd, p + "time_caption_embed.timestep_embedder.linear_1")),
                sd, p + "time_caption_embed.timestep_embedder.linear_2")
    pool = enc.mean(dim=1)  # all-ones encoder mask
    temb = te + linear(layer_norm(pool, sd, p + "time_caption_embed.caption_embedder.0", 1e-5), sd, p + "time_caption_embed.caption_embedder.1")
    for i in range(n_layers):
        x = dit_block(x, enc, temb, sd, f"{p}layers.{i}")
    scale = linear(F.silu(temb), sd, p + "norm_out.linear_1")
    x = F.layer_norm(x, (x.shape[-1],), None, None, 1e-6) * (1 + scale)[:, None, :]
    return linear(x, sd, p + "norm_out.linear_2")


def condition_tokens(sd, traj_latents, images_dp):
    """internvla_n1.py:364-381: cond_projector(latents) and the async memory tokens -> hidden_states [B, 32+n_query, 768]."""
    lat = linear(F.gelu(linear(traj_latents.float(), sd, "cond_projector.0"), approximate="tanh"), sd, "cond_projector.2")
    B, Fr = images_dp.shape[:2]
    img = images_dp.float().permute(0, 1, 4, 2, 3).flatten(0, 1)
    feat = dinov2.forward_tokens((img - RESNET_MEAN) / RESNET_STD, sd, "rgb_model.").reshape(B, Fr * 256, -1)
    mem = memory_encoder(feat, sd)
    tokens = qformer(torch.cat([feat, mem], dim=-1), sd)
    return torch.cat([tokens, lat], dim=1)


def generate_traj(sd, traj_latents, images_dp, x_init, num_inference_steps=10, num_sample_trajs=32, guidance_scale=1.0, use_async=True):
    """generate_traj, 'nextdit' [+ 'async'] (internvla_n1.py:349-432), looped over envs. traj_latents [B,n_query,3584];
    images_dp [B,2,224,224,3] in 0..1; x_init [B,S,T,3] (the reference's randn_tensor) -> latents [B,S,T,3].
    use_async=False: hidden_states = cond_projector(traj_latents) alone (:382-383); guidance_scale: the CFG weight of :425-427."""
    B = traj_latents.shape[0]
    S, T = x_init.shape[1:3]
    if use_async:
        hidden = condition_tokens(sd, traj_latents, images_dp)
    else:
        hidden = linear(F.gelu(linear(traj_latents.float(), sd, "cond_projector.0"), approximate="tanh"), sd, "cond_projector.2")
    sig = np.linspace(1.0, 1 / num_inference_steps, num_inference_steps).astype(np.float32)
    sigmas = torch.cat([torch.from_numpy(sig), torch.zeros(1)])
    timesteps = (torch.from_numpy(sig) * 1000.0)
    pos = sinusoidal_positional_encoding(T, 384)
    outs = []
    for b in range(B):
        hs = hidden[b:b + 1]
        z = torch.cat([torch.zeros_like(hs), hs], 0).repeat_interleave(S, dim=0)
        lat = x_init[b].float()
        for i, t in enumerate(timesteps):
            feat = linear(lat, sd, "action_encoder") + pos
            x_in = feat.repeat(2, 1, 1)
            ts = t.unsqueeze(0).expand(x_in.shape[0]).to(torch.long)
            pred = linear(traj_dit(x_in, ts, z, sd), sd, "action_decoder")
            u, c = pred.chunk(2)
            pred = u + guidance_scale * (c - u)
            lat = lat + (sigmas[i + 1] - sigmas[i]) * pred
        outs.append(lat)
    return torch.stack(outs)
