"""Oracle restatement of the DINOv2 ViT-S/14 encoder (DepthAnythingV2 `.pretrained`).  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/internnav/model/encoder/depth_anything/depth_anything_v2/:
  dinov2.py:399-411  DINOv2('vits'): img 518, patch 14, init_values 1.0 (LayerScale), interpolate_offset 0.1
  dinov2.py:340-351  vit_small: embed 384, depth 12, heads 6, mlp_ratio 4
  dinov2.py:180-232  interpolate_pos_encoding + prepare_tokens_with_masks
  dinov2.py:298-322  get_intermediate_layers(x)[0]: last block, final LayerNorm (eps 1e-6), cls token dropped
  dinov2_layers/patch_embed.py:108-164, attention.py:49-62, block.py:82-107, mlp.py, layer_scale.py
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .nn_ref import layer_norm, linear, sdpa

EMBED, DEPTH, HEADS, PATCH = 384, 12, 6, 14


def interpolate_pos_embed(pos_embed: torch.Tensor, w: int, h: int, offset: float = 0.1) -> torch.Tensor:
    """dinov2.py:180-211. pos_embed [1, 1+N, C] (N = 37*37 for img 518) -> [1, 1 + (w/14)*(h/14), C]; input independent."""
    N = pos_embed.shape[1] - 1
    npatch = (w // PATCH) * (h // PATCH)
    if npatch == N and w == h:
        return pos_embed
    pe = pos_embed.float()
    cls_pe, patch_pe = pe[:, 0], pe[:, 1:]
    dim = pe.shape[-1]
    w0, h0 = w // PATCH + offset, h // PATCH + offset
    s = math.sqrt(N)
    sx, sy = float(w0) / s, float(h0) / s
    patch_pe = F.interpolate(patch_pe.reshape(1, int(s), int(s), dim).permute(0, 3, 1, 2), scale_factor=(sx, sy),
                             mode="bicubic", antialias=False)
    assert int(w0) == patch_pe.shape[-2] and int(h0) == patch_pe.shape[-1]
    patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((cls_pe.unsqueeze(0), patch_pe), dim=1)


def block(x, sd, p):
    """block.py:82-107 (eval branch) with attention.py:49-62."""
    B, N, C = x.shape
    y = layer_norm(x, sd, p + ".norm1", 1e-6)
    qkv = linear(y, sd, p + ".attn.qkv").reshape(B, N, 3, HEADS, C // HEADS).permute(2, 0, 3, 1, 4)
    o = sdpa(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(B, N, C)
    x = x + linear(o, sd, p + ".attn.proj") * sd[p + ".ls1.gamma"]
    y = layer_norm(x, sd, p + ".norm2", 1e-6)
    y = linear(F.gelu(linear(y, sd, p + ".mlp.fc1")), sd, p + ".mlp.fc2")
    return x + y * sd[p + ".ls2.gamma"]


def forward_tokens(img: torch.Tensor, sd: dict, prefix: str = "") -> torch.Tensor:
    """get_intermediate_layers(img)[0]: img [n, 3, H, W] (already normalised) -> patch tokens [n, (H/14)(W/14), 384]."""
    p = prefix
    n, _, H, W = img.shape
    x = F.conv2d(img, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=PATCH)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((sd[p + "cls_token"].expand(n, -1, -1), x), dim=1)
    x = x + interpolate_pos_embed(sd[p + "pos_embed"], H, W)
    for i in range(DEPTH):
        x = block(x, sd, f"{p}blocks.{i}")
    x = layer_norm(x, sd, p + "norm", 1e-6)
    return x[:, 1:]
