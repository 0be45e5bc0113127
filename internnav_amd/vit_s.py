"""DINOv2 ViT-S/14 encoder (DepthAnythingV2 `.pretrained`) on the gfx950 op library.

Mirrors `DinoVisionTransformer.get_intermediate_layers(x)[0]` of the reference
(/root/reference/internnav/model/encoder/depth_anything/depth_anything_v2/dinov2.py:298-322; blocks
dinov2_layers/block.py:82-107, attention.py:49-62, patch_embed.py:151-164) with the caller's input normalisation
(navdp_backbone.py:155-181) fused into the im2col kernel. Host code only orders kernel launches; every FLOP runs in
libinternnav_amd.so. State-dict keys are the reference's, so a real `depth_anything_v2_vits.pth` loads unchanged.

Data layout in HBM (n = frames in the batch):
  patches  bf16 [n*256, 592]   im2col rows, k = c*196 + y*14 + x, 4 zero pad columns (K multiple of 8 for the MFMA GEMM)
  x        f32  [n*257, 384]   residual stream (cls + 256 patch tokens per frame), updated in place by GEMM epilogues
  h / att  bf16 [n*257, 384]   LayerNorm output / attention output (GEMM A operands)
  qkv      bf16 [n*257, 1152]  fused q|k|v, read by the attention kernel through strided views (no split copies)
  mlp      bf16 [n*257, 1536]
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import ops

D, DEPTH, HEADS, HD, PATCH = 384, 12, 6, 64, 14
KPAD = 592  # 3*14*14 = 588 padded to a multiple of 8


def interpolate_pos_embed(pos_embed: torch.Tensor, w: int, h: int, offset: float = 0.1) -> torch.Tensor:
    """Input-independent bicubic resampling of the 37x37 positional grid (dinov2.py:180-211); done once on the host at load."""
    N = pos_embed.shape[1] - 1
    if (w // PATCH) * (h // PATCH) == N and w == h:
        return pos_embed.float()
    pe = pos_embed.float()
    dim = pe.shape[-1]
    s = math.sqrt(N)
    w0, h0 = w // PATCH + offset, h // PATCH + offset
    grid = F.interpolate(pe[:, 1:].reshape(1, int(s), int(s), dim).permute(0, 3, 1, 2),
                         scale_factor=(float(w0) / s, float(h0) / s), mode="bicubic", antialias=False)
    assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
    return torch.cat((pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)


class VitWorkspace:
    """Activation buffers for up to `n_max` frames; shared by every ViT-S instance of a policy (they run back to back)."""

    def __init__(self, n_max: int, device):
        bf, f32 = torch.bfloat16, torch.float32
        self.n_max = n_max
        self.patches = torch.empty(n_max * 256, KPAD, dtype=bf, device=device)
        self.x = torch.empty(n_max * 257, D, dtype=f32, device=device)
        self.h = torch.empty(n_max * 257, D, dtype=bf, device=device)
        self.att = torch.empty(n_max * 257, D, dtype=bf, device=device)
        self.qkv = torch.empty(n_max * 257, 3 * D, dtype=bf, device=device)
        self.mlp = torch.empty(n_max * 257, 4 * D, dtype=bf, device=device)


class DinoV2Encoder:
    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, device, img_size: int = 224):
        bf, f32 = torch.bfloat16, torch.float32

        def w(k):
            return sd[prefix + k].to(device=device, dtype=bf).contiguous()

        def f(k):
            return sd[prefix + k].to(device=device, dtype=f32).contiguous()

        conv = sd[prefix + "patch_embed.proj.weight"].float().reshape(D, 3 * PATCH * PATCH)
        self.w_patch = F.pad(conv, (0, KPAD - conv.shape[1])).to(device=device, dtype=bf).contiguous()
        self.b_patch = f("patch_embed.proj.bias")
        pe = interpolate_pos_embed(sd[prefix + "pos_embed"].cpu(), img_size, img_size)
        self.pos = pe[0, 1:].to(device=device, dtype=f32).contiguous()                                   # [256, 384]
        self.cls_pos = (sd[prefix + "cls_token"].float().cpu()[0] + pe[0, :1]).to(device).contiguous()  # [1, 384]
        self.blocks = []
        for i in range(DEPTH):
            b = f"blocks.{i}."
            self.blocks.append(dict(
                n1w=f(b + "norm1.weight"), n1b=f(b + "norm1.bias"), qkv_w=w(b + "attn.qkv.weight"), qkv_b=f(b + "attn.qkv.bias"),
                proj_w=w(b + "attn.proj.weight"), proj_b=f(b + "attn.proj.bias"), ls1=f(b + "ls1.gamma"),
                n2w=f(b + "norm2.weight"), n2b=f(b + "norm2.bias"), fc1_w=w(b + "mlp.fc1.weight"), fc1_b=f(b + "mlp.fc1.bias"),
                fc2_w=w(b + "mlp.fc2.weight"), fc2_b=f(b + "mlp.fc2.bias"), ls2=f(b + "ls2.gamma")))
        self.norm_w, self.norm_b = f("norm.weight"), f("norm.bias")
        self.img_size = img_size
        self.grid = img_size // PATCH
        assert self.grid * self.grid == 256, "workspace layout assumes 224x224 frames (16x16 patches)"

    def forward(self, frames: torch.Tensor, ws: VitWorkspace, out: Optional[torch.Tensor], out_map=None,
                pos: Optional[torch.Tensor] = None, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), extra_outputs=()) -> torch.Tensor:
        """frames [n, 224, 224, C] (C = 3, or 1 replicated to 3 channels; f32 or bf16) -> bf16 patch tokens written to
        `out` rows out_map(i*256 + p) (+ pos[(i*256+p) % len(pos)]); the cls token is dropped as in the reference.
        extra_outputs: further (out_bf16 | None, out_f32 | None, out_map, pos) destinations of the same final LayerNorm
        (e.g. tokens + a positional table as the fp32 stream of the next module)."""
        n = frames.shape[0]
        assert n <= ws.n_max and frames.shape[1] == frames.shape[2] == self.img_size
        T = 257
        patches, x, h, att, qkv, mlp = (ws.patches[: n * 256], ws.x[: n * T], ws.h[: n * T], ws.att[: n * T],
                                        ws.qkv[: n * T], ws.mlp[: n * T])
        ops.patchify(frames, patches, mean, std, PATCH)
        x3 = x.view(n, T, D)
        ops.linear(patches.view(n, 256, KPAD), self.w_patch, bias=self.b_patch, residual=self.pos, out=x3[:, 1:, :], batched=True)
        ops.embed3(None, None, None, out=x, pos=self.cls_pos, rows=n, out_map=(1, T, 0))
        qkv5 = qkv.view(n, T, 3, HEADS, HD)
        for b in self.blocks:
            ops.norm(x, b["n1w"], b["n1b"], eps=1e-6, out=h)
            ops.linear(h, b["qkv_w"], bias=b["qkv_b"], out=qkv)
            ops.attention(qkv5[:, :, 0], qkv5[:, :, 1], qkv5[:, :, 2], out=att.view(n, T, HEADS, HD))
            ops.linear(att, b["proj_w"], bias=b["proj_b"], colscale=b["ls1"], residual=x, out=x)
            ops.norm(x, b["n2w"], b["n2b"], eps=1e-6, out=h)
            ops.linear(h, b["fc1_w"], bias=b["fc1_b"], act="gelu", out=mlp)
            ops.linear(mlp, b["fc2_w"], bias=b["fc2_b"], colscale=b["ls2"], residual=x, out=x)
        if out is not None:
            ops.norm(x, self.norm_w, self.norm_b, eps=1e-6, out=out, rows=n * 256, in_map=(256, T, 1), out_map=out_map, pos=pos)
        for (o16, o32, omap, opos) in extra_outputs:
            ops.norm(x, self.norm_w, self.norm_b, eps=1e-6, out=o16, out32=o32, rows=n * 256, in_map=(256, T, 1), out_map=omap, pos=opos)
        return out
