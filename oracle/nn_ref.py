"""Oracle primitives (fp32, CPU): explicit restatements of the torch.nn building blocks the reference composes.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). `sd` is always a flat state-dict with the reference's key names;
`p` is the key prefix of the sub-module being evaluated.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def linear(x, sd, p):
    """nn.Linear: y = x W^T + b."""
    b = sd.get(p + ".bias")
    return F.linear(x, sd[p + ".weight"], b)


def layer_norm(x, sd, p, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd.get(p + ".bias"), eps)


def rms_norm(x, w, eps):
    """diffusers RMSNorm / Qwen2_5_VLRMSNorm: x * rsqrt(mean(x^2) + eps) * w (statistics in fp32)."""
    v = x.float().pow(2).mean(-1, keepdim=True)
    y = x.float() * torch.rsqrt(v + eps)
    return y * w if w is not None else y


def sdpa(q, k, v, mask=None, scale=None):
    """softmax(q k^T * scale + mask) v with q [B,H,Lq,D], k/v [B,H,Lk,D]; mask additive float or bool (True = keep)."""
    d = q.shape[-1]
    s = (q @ k.transpose(-1, -2)) * (scale if scale is not None else 1.0 / math.sqrt(d))
    if mask is not None:
        if mask.dtype == torch.bool:
            s = s.masked_fill(~mask, float("-inf"))
        else:
            s = s + mask
    return torch.softmax(s, dim=-1) @ v


def mha(query, key, value, sd, p, nhead, attn_mask=None):
    """nn.MultiheadAttention(batch_first=True) forward, eval mode: packed in_proj, per-head SDPA, out_proj."""
    E = query.shape[-1]
    W, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(query, W[:E], b[:E])
    k = F.linear(key, W[E:2 * E], b[E:2 * E])
    v = F.linear(value, W[2 * E:], b[2 * E:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    hd = E // nhead
    q = q.view(B, Lq, nhead, hd).transpose(1, 2)
    k = k.view(B, Lk, nhead, hd).transpose(1, 2)
    v = v.view(B, Lk, nhead, hd).transpose(1, 2)
    o = sdpa(q, k, v, attn_mask)
    o = o.transpose(1, 2).reshape(B, Lq, E)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _act(name):
    return {"relu": F.relu, "gelu": F.gelu}[name]


def decoder_layer(x, mem, sd, p, nhead, norm_first, act, tgt_mask=None, memory_mask=None, eps=1e-5):
    """nn.TransformerDecoderLayer(batch_first=True) forward (torch/nn/modules/transformer.py), dropout off."""
    def sa(t):
        return mha(t, t, t, sd, p + ".self_attn", nhead, tgt_mask)

    def ca(t):
        return mha(t, mem, mem, sd, p + ".multihead_attn", nhead, memory_mask)

    def ff(t):
        return linear(_act(act)(linear(t, sd, p + ".linear1")), sd, p + ".linear2")

    if norm_first:
        x = x + sa(layer_norm(x, sd, p + ".norm1", eps))
        x = x + ca(layer_norm(x, sd, p + ".norm2", eps))
        x = x + ff(layer_norm(x, sd, p + ".norm3", eps))
    else:
        x = layer_norm(x + sa(x), sd, p + ".norm1", eps)
        x = layer_norm(x + ca(x), sd, p + ".norm2", eps)
        x = layer_norm(x + ff(x), sd, p + ".norm3", eps)
    return x


def encoder_layer(x, sd, p, nhead, act="relu", eps=1e-5):
    """nn.TransformerEncoderLayer(batch_first=True, norm_first=False) forward, dropout off."""
    x = layer_norm(x + mha(x, x, x, sd, p + ".self_attn", nhead), sd, p + ".norm1", eps)
    x = layer_norm(x + linear(_act(act)(linear(x, sd, p + ".linear1")), sd, p + ".linear2"), sd, p + ".norm2", eps)
    return x


def causal_mask(T):
    """the reference's tgt_mask (navdp_policy.py:122-128, navdp.py:82-88): 0 on/below the diagonal, -inf above."""
    m = torch.full((T, T), float("-inf"))
    return torch.triu(m, diagonal=1)


def sinusoidal_pos_emb(x, dim):
    """SinusoidalPosEmb (navdp_backbone.py:9-21 == diffusion_policy positional_embedding.py:5-17)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float32) * -e)
    e = x[:, None].float() * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)
