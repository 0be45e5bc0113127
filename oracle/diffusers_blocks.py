"""Restatement of the diffusers==0.33.1 building blocks the reference's NextDiT imports.  TEST INFRASTRUCTURE ONLY.

diffusers is an un-vendored third-party dependency of the reference (requirements/internvla_n1.txt:3; import sites
internvla_n1/nextdit_traj.py:19-34, nextdit_crossattn_traj.py:4, internvla_n1.py:7-8) and is not installed in this
image. These nn.Modules restate the published algorithm of each class so that the reference's OWN in-tree wiring
(`LuminaNextDiTBlock`, `LuminaNextDiT2DModel`, `NextDiTCrossAttn`) can be executed here by `oracle/make_golden.py`
under a fake `diffusers` package. They pin the in-tree wiring only; the block internals stay "parity unpinned"
 until a diffusers wheel or a real checkpoint key list is available.
"""
from __future__ import annotations

import math
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


class RMSNorm(nn.Module):
    def __init__(self, dim, eps, elementwise_affine=True, bias=False):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim)) if elementwise_affine else None

    def forward(self, x):
        dt = x.dtype
        v = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        x = x * torch.rsqrt(v + self.eps)
        if self.weight is not None:
            if self.weight.dtype in (torch.float16, torch.bfloat16):
                x = x.to(self.weight.dtype)
            x = x * self.weight
        else:
            x = x.to(dt)
        return x


class FP32SiLU(nn.Module):
    def forward(self, x):
        return F.silu(x.float(), inplace=False).to(x.dtype)


class LuminaRMSNormZero(nn.Module):
    def __init__(self, embedding_dim, norm_eps, norm_elementwise_affine):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(min(embedding_dim, 1024), 4 * embedding_dim, bias=True)
        self.norm = RMSNorm(embedding_dim, eps=norm_eps)

    def forward(self, x, emb=None):
        emb = self.linear(self.silu(emb))
        scale_msa, gate_msa, scale_mlp, gate_mlp = emb.chunk(4, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None])
        return x, gate_msa, scale_mlp, gate_mlp


class LuminaLayerNormContinuous(nn.Module):
    def __init__(self, embedding_dim, conditioning_embedding_dim, elementwise_affine=True, eps=1e-5, bias=True,
                 norm_type="layer_norm", out_dim=None):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear_1 = nn.Linear(conditioning_embedding_dim, embedding_dim, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, eps, elementwise_affine, bias)
        self.linear_2 = nn.Linear(embedding_dim, out_dim, bias=bias) if out_dim is not None else None

    def forward(self, x, conditioning_embedding):
        emb = self.linear_1(self.silu(conditioning_embedding).to(x.dtype))
        x = self.norm(x) * (1 + emb)[:, None, :]
        if self.linear_2 is not None:
            x = self.linear_2(x)
        return x


# Which LuminaFeedForward the in-tree `LuminaNextDiTBlock` is built on decides the width of the trajectory DiT's FFN, and it changed
# between diffusers releases (restated from the published sources - the package is absent here, DESIGN.md 2 "FFN width"):
#   * up to 0.32.x  the CLASS shrinks its argument: inner_dim = int(2 * inner_dim / 3) -> the reference's `inner_dim=4 * dim`
#                   (nextdit_traj.py:102-107) gives 256 * ceil(1024 / 256) = 1024;
#   * from 0.33.0   (the Lumina2 refactor) the factor moved out of the class into diffusers' own LuminaNextDiTBlock call site
#                   (`inner_dim=int(4 * 2 * dim / 3)`), so the reference's unchanged in-tree call gives 256 * ceil(1536 / 256) = 1536.
# The reference pins diffusers==0.33.1 (requirements/internvla_n1.txt:3): the pinned reading is 1536 and is the default here. The other
# convention stays selectable so both widths are pinned through the reference's own block wiring (tests/golden/n1_nextdit*.pt).
LEGACY_TWO_THIRDS = False


class ffn_convention:
    """`with ffn_convention(legacy_two_thirds=True): ...` builds LuminaFeedForward as diffusers <= 0.32 did (width 1024 for dim 384)."""

    def __init__(self, legacy_two_thirds: bool):
        self.legacy = bool(legacy_two_thirds)

    def __enter__(self):
        global LEGACY_TWO_THIRDS
        self.prev, LEGACY_TWO_THIRDS = LEGACY_TWO_THIRDS, self.legacy
        return self

    def __exit__(self, *exc):
        global LEGACY_TWO_THIRDS
        LEGACY_TWO_THIRDS = self.prev
        return False


def lumina_ffn_width(inner_dim: int, multiple_of: int = 256, ffn_dim_multiplier=None, legacy_two_thirds: bool = False) -> int:
    if legacy_two_thirds:
        inner_dim = int(2 * inner_dim / 3)
    if ffn_dim_multiplier is not None:
        inner_dim = int(ffn_dim_multiplier * inner_dim)
    return multiple_of * ((inner_dim + multiple_of - 1) // multiple_of)


class LuminaFeedForward(nn.Module):
    def __init__(self, dim, inner_dim, multiple_of=256, ffn_dim_multiplier=None, legacy_two_thirds=None):
        super().__init__()
        legacy = LEGACY_TWO_THIRDS if legacy_two_thirds is None else legacy_two_thirds
        inner_dim = lumina_ffn_width(inner_dim, multiple_of, ffn_dim_multiplier, legacy)
        self.linear_1 = nn.Linear(dim, inner_dim, bias=False)
        self.linear_2 = nn.Linear(inner_dim, dim, bias=False)
        self.linear_3 = nn.Linear(dim, inner_dim, bias=False)
        self.silu = FP32SiLU()

    def forward(self, x):
        return self.linear_2(self.silu(self.linear_1(x)) * self.linear_3(x))


class PixArtAlphaTextProjection(nn.Module):
    def __init__(self, in_features, hidden_size, out_features=None, act_fn="gelu_tanh"):
        super().__init__()
        out_features = out_features or hidden_size
        self.linear_1 = nn.Linear(in_features, hidden_size, bias=True)
        self.act_1 = nn.GELU(approximate="tanh")
        self.linear_2 = nn.Linear(hidden_size, out_features, bias=True)

    def forward(self, caption):
        return self.linear_2(self.act_1(self.linear_1(caption)))


def get_timestep_embedding(timesteps, dim, flip_sin_to_cos=False, downscale_freq_shift=1.0, scale=1.0, max_period=10000):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift, scale=1):
        super().__init__()
        self.num_channels, self.flip, self.shift, self.scale = num_channels, flip_sin_to_cos, downscale_freq_shift, scale

    def forward(self, t):
        return get_timestep_embedding(t, self.num_channels, self.flip, self.shift, self.scale)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim, True)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim, True)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


class LuminaCombinedTimestepCaptionEmbedding(nn.Module):
    def __init__(self, hidden_size=4096, cross_attention_dim=2048, frequency_embedding_size=256):
        super().__init__()
        self.time_proj = Timesteps(frequency_embedding_size, flip_sin_to_cos=True, downscale_freq_shift=0.0)
        self.timestep_embedder = TimestepEmbedding(frequency_embedding_size, hidden_size)
        self.caption_embedder = nn.Sequential(nn.LayerNorm(cross_attention_dim),
                                              nn.Linear(cross_attention_dim, hidden_size, bias=True))

    def forward(self, timestep, caption_feat, caption_mask):
        time_freq = self.time_proj(timestep)
        time_embed = self.timestep_embedder(time_freq.to(dtype=caption_feat.dtype))
        m = caption_mask.float().unsqueeze(-1)
        pool = (caption_feat * m).sum(dim=1) / m.sum(dim=1)
        pool = pool.to(caption_feat)
        return time_embed + self.caption_embedder(pool)


class LuminaPatchEmbed(nn.Module):
    """constructed by LuminaNextDiT2DModel (nextdit_traj.py:253-255) but never called on the trajectory path."""

    def __init__(self, patch_size=2, in_channels=4, embed_dim=768, bias=True):
        super().__init__()
        self.patch_size = patch_size
        self.proj = nn.Linear(patch_size * patch_size * in_channels, embed_dim, bias=bias)


class LuminaAttnProcessor2_0:
    def __call__(self, attn, hidden_states, encoder_hidden_states, attention_mask=None, query_rotary_emb=None,
                 key_rotary_emb=None, base_sequence_length=None):
        assert query_rotary_emb is None and key_rotary_emb is None  # nextdit_crossattn_traj.py:92 passes None
        B, L, _ = hidden_states.shape
        q = attn.to_q(hidden_states)
        k = attn.to_k(encoder_hidden_states)
        v = attn.to_v(encoder_hidden_states)
        qd, inner = q.shape[-1], k.shape[-1]
        hd = qd // attn.heads
        dt = q.dtype
        kvh = inner // hd
        if attn.norm_q is not None:
            q = attn.norm_q(q)
        if attn.norm_k is not None:
            k = attn.norm_k(k)
        q = q.view(B, -1, attn.heads, hd).to(dt)
        k = k.view(B, -1, kvh, hd).to(dt)
        v = v.view(B, -1, kvh, hd)
        rep = attn.heads // kvh
        if rep >= 1:
            k = k.unsqueeze(3).repeat(1, 1, 1, rep, 1).flatten(2, 3)
            v = v.unsqueeze(3).repeat(1, 1, 1, rep, 1).flatten(2, 3)
        m = attention_mask.bool().view(B, 1, 1, -1).expand(-1, attn.heads, L, -1)
        o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=m, scale=None)
        return o.transpose(1, 2).to(dt)


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, kv_heads=None, dim_head=64, bias=False,
                 qk_norm=None, eps=1e-5, out_bias=True, processor=None, **_):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.inner_kv_dim = self.inner_dim if kv_heads is None else dim_head * kv_heads
        self.heads = heads
        self.scale = dim_head ** -0.5
        cad = cross_attention_dim if cross_attention_dim is not None else query_dim
        if qk_norm == "layer_norm_across_heads":
            self.norm_q = nn.LayerNorm(dim_head * heads, eps=eps)
            self.norm_k = nn.LayerNorm(dim_head * (kv_heads or heads), eps=eps)
        else:
            assert qk_norm is None
            self.norm_q = self.norm_k = None
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(cad, self.inner_kv_dim, bias=bias)
        self.to_v = nn.Linear(cad, self.inner_kv_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(0.0)])
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states, attention_mask=attention_mask, **kw)


def get_2d_rotary_pos_embed_lumina(embed_dim, len_h, len_w, linear_factor=1.0, ntk_factor=1.0):
    """computed at nextdit_crossattn_traj.py:80-84 and never used (image_rotary_emb=None at :92)."""
    return None


def install_fake_diffusers(sch) -> None:
    """Register a minimal `diffusers` package in sys.modules (only for oracle/make_golden.py)."""
    def mod(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    d = mod("diffusers")
    s = mod("diffusers.schedulers")
    s.FlowMatchEulerDiscreteScheduler = sch.FlowMatchEulerDiscreteScheduler
    s.DDPMScheduler = sch.DDPMScheduler
    sd = mod("diffusers.schedulers.scheduling_ddpm")
    sd.DDPMScheduler = sch.DDPMScheduler
    d.schedulers = s
    cu = mod("diffusers.configuration_utils")

    class ConfigMixin:
        pass

    def register_to_config(fn):
        return fn

    cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config
    mod("diffusers.models")
    a = mod("diffusers.models.attention")
    a.LuminaFeedForward = LuminaFeedForward
    ap = mod("diffusers.models.attention_processor")
    ap.Attention, ap.LuminaAttnProcessor2_0 = Attention, LuminaAttnProcessor2_0
    e = mod("diffusers.models.embeddings")
    e.LuminaCombinedTimestepCaptionEmbedding = LuminaCombinedTimestepCaptionEmbedding
    e.LuminaPatchEmbed, e.PixArtAlphaTextProjection = LuminaPatchEmbed, PixArtAlphaTextProjection
    e.get_2d_rotary_pos_embed_lumina = get_2d_rotary_pos_embed_lumina
    mo = mod("diffusers.models.modeling_outputs")

    class Transformer2DModelOutput:
        def __init__(self, sample):
            self.sample = sample

    mo.Transformer2DModelOutput = Transformer2DModelOutput
    mu = mod("diffusers.models.modeling_utils")

    class ModelMixin(nn.Module):
        def enable_gradient_checkpointing(self):
            self.gradient_checkpointing = True

    mu.ModelMixin = ModelMixin
    n = mod("diffusers.models.normalization")
    n.LuminaLayerNormContinuous, n.LuminaRMSNormZero, n.RMSNorm = LuminaLayerNormContinuous, LuminaRMSNormZero, RMSNorm
    u = mod("diffusers.utils")
    u.is_torch_version = lambda op, v: True
    import logging as _logging

    u.logging = types.SimpleNamespace(get_logger=_logging.getLogger)
    tu = mod("diffusers.utils.torch_utils")
    tu.randn_tensor = lambda shape, generator=None, device=None, dtype=None: torch.randn(shape, generator=generator,
                                                                                       device=device, dtype=dtype)
