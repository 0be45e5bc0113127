"""Seeded synthetic state-dicts (the reference's parameter names and shapes) and synthetic inputs.

No checkpoint exists offline, so parity tests, smoke() and bench.py run on seeded random weights at the
true architecture shapes ("data": "synthetic" in the bench line). The key tables below are written from the reference
constructors (cited per function) and are checked against the real modules' `state_dict()` by `oracle/make_golden.py`
(strict key/shape comparison). Host-side only (torch CPU generators); the engines upload the tensors themselves.

Initialisation is "active" rather than the reference's N(0, 0.02) default: every Linear has unit gain
(std = fan_in^-0.5), biases / norm affine terms / LayerScale are perturbed, so that every branch of every block
contributes O(1) to the residual stream and a wrong kernel cannot hide under the tolerance. Each tensor is drawn from
its own generator seeded by crc32(key) ^ seed (order independent) and rounded to bf16-representable values, so the
fp32 oracle and the bf16 HIP engine consume bit-identical parameters.
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import torch

Spec = Dict[str, Tuple[tuple, str]]


def _draw(key: str, shape, kind: str, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    if kind == "w":  # [out, in, ...] unit-gain linear / conv weight
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        t = r * fan_in ** -0.5
    elif kind == "w_small":  # projections whose input is not normalised (K/V of raw memory, heads)
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        t = r * 0.5 * fan_in ** -0.5
    elif kind == "b":
        t = r * 0.1
    elif kind == "ln_w":
        t = 1.0 + 0.1 * r
    elif kind == "ln_b":
        t = 0.1 * r
    elif kind == "emb":
        t = 0.3 * r
    elif kind == "gamma":
        t = 0.5 + 0.1 * r
    elif kind == "gate":
        t = 0.5 * r
    elif kind == "latent":
        t = r
    else:
        raise ValueError(kind)
    return t.to(torch.bfloat16).to(torch.float32)


class LazyDeviceWeights:
    """Mapping key -> tensor that draws each parameter directly on the device when it is first read (bf16-representable values,
    same scale rules as `_draw`, a per-key device generator). Used by bench.py for the 7B-parameter System-2: the state dict
    never exists on the host (15 GB), and each tensor is released as soon as the engine has repacked it."""

    _SCALE = {"b": 0.1, "ln_b": 0.1, "emb": 0.3, "gate": 0.5, "latent": 1.0}

    def __init__(self, spec: Spec, device, seed: int = 0, prefixes=("",)):
        self.spec, self.device, self.seed = spec, torch.device(device), seed

    def __contains__(self, key):
        return key in self.spec

    def keys(self):
        return self.spec.keys()

    def shape_of(self, key: str):
        return tuple(self.spec[key][0])

    def __getitem__(self, key: str) -> torch.Tensor:
        shape, kind = self.spec[key]
        g = torch.Generator(device=self.device).manual_seed((zlib.crc32(key.encode()) ^ (self.seed * 0x9E3779B1)) & 0x7FFFFFFF)
        r = torch.randn(shape, generator=g, dtype=torch.float32, device=self.device)
        if kind in ("w", "w_small"):
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            r.mul_((0.5 if kind == "w_small" else 1.0) * fan_in ** -0.5)
        elif kind == "ln_w":
            r.mul_(0.1).add_(1.0)
        elif kind == "gamma":
            r.mul_(0.1).add_(0.5)
        else:
            r.mul_(self._SCALE[kind])
        return r.to(torch.bfloat16)


_HASH_KIND = {"w": (None, 0.0), "w_small": (None, 0.0), "b": (0.1, 0.0), "ln_w": (0.1, 1.0), "ln_b": (0.1, 0.0), "emb": (0.3, 0.0),
              "gamma": (0.1, 0.5), "gate": (0.5, 0.0), "latent": (1.0, 0.0)}


def hash_uniform(n: int, seed: int, device="cpu", chunk: int = 1 << 24) -> torch.Tensor:
    """n fp32 values in [-0.5, 0.5) from a counter hash of (seed, index): integer arithmetic only (32-bit mixes carried in int64, every
    product < 2^63), so torch evaluates it to the SAME bits on the CPU and on the GPU - unlike torch.randn, whose CPU and device
    generators are different streams. Used where a fixture generated in the build container (CPU) must meet weights created on the GPU."""
    out = torch.empty(n, dtype=torch.float32, device=device)
    M = 0xFFFFFFFF
    for s0 in range(0, n, chunk):
        m = min(chunk, n - s0)
        x = torch.arange(s0, s0 + m, dtype=torch.int64, device=device)
        x = (x * 2 + 1 + (seed & M)) & M
        x = ((x ^ (x >> 16)) * 0x45D9F3B) & M
        x = ((x ^ (x >> 15)) * 0x2C1B3C6D) & M
        x = ((x ^ (x >> 16)) * 0x297A2D39) & M
        x = x ^ (x >> 15)
        out[s0:s0 + m] = (x >> 8).to(torch.float32).mul_(2.0 ** -24).sub_(0.5)
    return out


class HashWeights:
    """Mapping key -> bf16 tensor drawn lazily from `hash_uniform` (uniform with the same standard deviation per kind as `_draw`):
    bit-identical on cpu and cuda, seconds for the 7.6 B-parameter System-2 on the GPU. The full-configuration parity fixture
    (oracle/make_golden_full.py, CPU) and its GPU test (tests/test_qwen_full_gpu.py) both build their weights here."""

    def __init__(self, spec: Spec, seed: int = 0, device="cpu"):
        self.spec, self.seed, self.device = spec, seed, torch.device(device)

    def __contains__(self, key):
        return key in self.spec

    def keys(self):
        return self.spec.keys()

    def __getitem__(self, key: str) -> torch.Tensor:
        shape, kind = self.spec[key]
        n = 1
        for d in shape:
            n *= d
        std, mean = _HASH_KIND[kind]
        if std is None:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            std = (0.5 if kind == "w_small" else 1.0) * fan_in ** -0.5
        u = hash_uniform(n, (zlib.crc32(key.encode()) ^ (self.seed * 0x9E3779B1)) & 0x7FFFFFFF, self.device)
        u.mul_(float(torch.tensor(12.0 ** 0.5 * std, dtype=torch.float32)))
        if mean:
            u.add_(mean)
        return u.to(torch.bfloat16).view(shape)


OUTLIER_CHANNELS = (143, 647, 1151, 1879, 2495, 3083)
OUTLIER_SCALES = (128, 256, 512, 1024, 256, 512)
OUTLIER_OTHERS_GAIN = 2


class OutlierHashWeights(HashWeights):
    """HashWeights of the System-2 decoder with MASSIVE-ACTIVATION channels, as released Qwen2.5 checkpoints carry them (every
    other parity run uses N(0, 0.02)-like weights, whose residual stream has no outliers). Six rows of layer 0's down projection are scaled
    by 128 ... 1024, so from layer 0 on six channels of the residual stream sit at 60 ... 500 where the others are O(1) (x100 - x1000), for
    every token and through all later layers. As in the real models the norm gains absorb them: in every RMSNorm behind layer 0 the gain of an
    outlier channel is multiplied by 64 / scale (it enters the next GEMM at ~2.5x a normal channel) and all other gains by 2. The outliers
    inflate the row RMS ~13x, so the other channels enter the later layers at ~0.15-0.3 of their usual size: the later layers are attenuated,
    deliberately - with gains of 3 and more the random-weight network turns chaotic (bf16 PyTorch itself drifts to O(1) relative error within
    ten layers, measured) and is useless as a yardstick; with 2 its bf16 error stays at the percent level over the 28 layers.
    All factors are powers of two: exact in bf16, so the CPU fixture (oracle/make_golden_full.py --outliers) and the GPU test draw identical bits."""

    def __getitem__(self, key: str) -> torch.Tensor:
        w = super().__getitem__(key)
        ch = torch.tensor(OUTLIER_CHANNELS, device=w.device)
        sc = torch.tensor(OUTLIER_SCALES, device=w.device, dtype=torch.float32)
        if key == "model.layers.0.mlp.down_proj.weight":
            w[ch] = (w[ch].float() * sc[:, None]).to(torch.bfloat16)
        elif key == "model.norm.weight" or (key.startswith("model.layers.") and key.endswith("layernorm.weight") and int(key.split(".")[2]) >= 1):
            g = torch.full(w.shape, float(OUTLIER_OTHERS_GAIN), device=w.device)
            g[ch] = 64.0 / sc
            w = (w.float() * g).to(torch.bfloat16)
        return w


def materialize(spec: Spec, seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: _draw(k, shape, kind, seed) for k, (shape, kind) in spec.items()}


# ------------------------------------------------------------------------------------------------------ spec builders
def _lin(spec: Spec, p: str, out_f: int, in_f: int, bias: bool = True, kind: str = "w"):
    spec[p + ".weight"] = ((out_f, in_f), kind)
    if bias:
        spec[p + ".bias"] = ((out_f,), "b")


def _ln(spec: Spec, p: str, d: int, bias: bool = True):
    spec[p + ".weight"] = ((d,), "ln_w")
    if bias:
        spec[p + ".bias"] = ((d,), "ln_b")


def _mha(spec: Spec, p: str, d: int):
    """nn.MultiheadAttention parameters."""
    spec[p + ".in_proj_weight"] = ((3 * d, d), "w")
    spec[p + ".in_proj_bias"] = ((3 * d,), "b")
    _lin(spec, p + ".out_proj", d, d)


def dinov2_vits_spec(p: str = "") -> Spec:
    """DINOv2('vits') (depth_anything_v2/dinov2.py:399-411): 22.06 M parameters."""
    s: Spec = {}
    D = 384
    s[p + "cls_token"] = ((1, 1, D), "emb")
    s[p + "pos_embed"] = ((1, 37 * 37 + 1, D), "emb")
    s[p + "mask_token"] = ((1, D), "emb")
    s[p + "patch_embed.proj.weight"] = ((D, 3, 14, 14), "w")
    s[p + "patch_embed.proj.bias"] = ((D,), "b")
    for i in range(12):
        b = f"{p}blocks.{i}"
        _ln(s, b + ".norm1", D)
        _lin(s, b + ".attn.qkv", 3 * D, D)
        _lin(s, b + ".attn.proj", D, D)
        s[b + ".ls1.gamma"] = ((D,), "gamma")
        _ln(s, b + ".norm2", D)
        _lin(s, b + ".mlp.fc1", 4 * D, D)
        _lin(s, b + ".mlp.fc2", D, 4 * D)
        s[b + ".ls2.gamma"] = ((D,), "gamma")
    _ln(s, p + "norm", D)
    return s


def decoder_layer_spec(p: str, d: int, ffn: int) -> Spec:
    """nn.TransformerDecoderLayer parameters."""
    s: Spec = {}
    _mha(s, p + ".self_attn", d)
    _mha(s, p + ".multihead_attn", d)
    _lin(s, p + ".linear1", ffn, d)
    _lin(s, p + ".linear2", d, ffn)
    for n in ("norm1", "norm2", "norm3"):
        _ln(s, f"{p}.{n}", d)
    return s


def encoder_layer_spec(p: str, d: int, ffn: int) -> Spec:
    s: Spec = {}
    _mha(s, p + ".self_attn", d)
    _lin(s, p + ".linear1", ffn, d)
    _lin(s, p + ".linear2", d, ffn)
    for n in ("norm1", "norm2"):
        _ln(s, f"{p}.{n}", d)
    return s


def rgbd_backbone_spec(p: str, memory_size: int, token_dim: int, dat: bool) -> Spec:
    """RGBDBackbone (navdp_backbone.py:205-246) or, dat=True, DAT_RGBD_Patch_Backbone version>0 (:102-149)."""
    s: Spec = {}
    s.update(dinov2_vits_spec(p + "rgb_model."))
    s.update(dinov2_vits_spec(p + "depth_model."))
    if dat:
        s[p + "former_query.weight"] = ((memory_size * 16, 384), "emb")
        s[p + "former_pe.weight"] = ((memory_size * 2 * 256, 384), "emb")
    else:
        s[p + "former_query.position_embedding.weight"] = ((memory_size * 16, 384), "emb")
        s[p + "former_pe.position_embedding.weight"] = (((memory_size + 1) * 256, 384), "emb")
    for i in range(2):
        s.update(decoder_layer_spec(f"{p}former_net.layers.{i}", 384, 2048))
    _lin(s, p + "project_layer", token_dim, 384)
    return s


NAVDPNET_CFG = dict(image_size=224, memory_size=8, predict_size=24, temporal_depth=16, heads=8, token_dim=384,
                    num_train_timesteps=10, sample_num=32)
"""NavDPNet hyper-parameters of BASELINE config #2 (scripts/train/base_train/configs/navdp.py:58-63; DDPM 10 steps navdp_policy.py:119-121)."""

N1_NAVDP_CFG = dict(image_size=224, memory_size=2, predict_size=32, temporal_depth=16, heads=8, token_dim=384,
                    vlm_token_dim=3584, num_train_timesteps=20, sample_num=32)
"""NavDP_Policy_DPT_CriticSum_DAT defaults (internvla_n1/navdp.py:17-34) as built by build_navdp (internvla_n1_arch.py:10-15)."""


def navdpnet_spec(cfg=NAVDPNET_CFG) -> Spec:
    """NavDPNet parameters used by the point-goal inference path (navdp_policy.py:67-133); the image/pixel goal encoders
    and aux heads exist in the reference module but are not on this path and are left at their own initialisation."""
    D, M, T = cfg["token_dim"], cfg["memory_size"], cfg["predict_size"]
    s: Spec = {}
    s.update(rgbd_backbone_spec("rgbd_encoder.", M, D, dat=False))
    _lin(s, "point_encoder", D, 3)
    for i in range(cfg["temporal_depth"]):
        s.update(decoder_layer_spec(f"decoder.layers.{i}", D, 4 * D))
    _lin(s, "input_embed", D, 3)
    s["cond_pos_embed.position_embedding.weight"] = ((M * 16 + 4, D), "emb")
    s["out_pos_embed.position_embedding.weight"] = ((T, D), "emb")
    _ln(s, "layernorm", D)
    _lin(s, "action_head", 3, D, kind="w_small")
    _lin(s, "critic_head", 1, D)
    return s


def n1_navdp_spec(cfg=N1_NAVDP_CFG) -> Spec:
    """NavDP_Policy_DPT_CriticSum_DAT parameters on the navdp_async inference path (internvla_n1/navdp.py:52-108)."""
    D, M, T, V = cfg["token_dim"], cfg["memory_size"], cfg["predict_size"], cfg["vlm_token_dim"]
    s: Spec = {}
    s.update(rgbd_backbone_spec("rgbd_encoder.", M, D, dat=True))
    for i in range(cfg["temporal_depth"]):
        s.update(decoder_layer_spec(f"decoder.layers.{i}", D, 4 * D))
    _lin(s, "input_embed", D, 3)
    s["cond_pos_embed"] = ((1, M * 16 + 2, D), "emb")
    s["out_pos_embed"] = ((1, T, D), "emb")
    _ln(s, "layernorm", D)
    _lin(s, "action_head", 3, D, kind="w_small")
    _lin(s, "vlm_embed_mlp.0", V // 4, V)
    _lin(s, "vlm_embed_mlp.2", V // 8, V // 4)
    _lin(s, "vlm_embed_mlp.4", D, V // 8)
    s["goal_compressor.target_embedding.weight"] = ((1, D), "emb")
    s["goal_compressor.token_positional_encoding.position_embedding.weight"] = ((5000, D), "emb")
    s["goal_compressor.query_positional_encoding.position_embedding.weight"] = ((5000, D), "emb")
    _mha(s, "goal_compressor.cross_attention", D)
    return s


def lumina_ffn_width(dim: int, multiple_of: int = 256, ffn_dim_multiplier=None, legacy_two_thirds: bool = False) -> int:
    """width of `LuminaFeedForward(dim, inner_dim=4 * dim, multiple_of, ffn_dim_multiplier)` as the reference's in-tree block builds it
    (nextdit_traj.py:102-107). diffusers >= 0.33.0 (the reference pins 0.33.1, requirements/internvla_n1.txt:3) takes inner_dim as given:
    256 * ceil(1536 / 256) = 1536 for dim 384; diffusers <= 0.32 shrank it inside the class (int(2 * inner_dim / 3) -> 1024). A checkpoint's
    own `feed_forward.linear_1.weight` shape is the authority (`n1_nextdit_cfg_from_weights`); this formula only sizes synthetic weights."""
    inner = 4 * dim
    if legacy_two_thirds:
        inner = int(2 * inner / 3)
    if ffn_dim_multiplier is not None:
        inner = int(ffn_dim_multiplier * inner)
    return multiple_of * ((inner + multiple_of - 1) // multiple_of)


N1_NEXTDIT_CFG = dict(n_query=4, vlm_token_dim=3584, latent_dim=768, dit_dim=384, dit_layers=12, dit_heads=6, dit_ffn=lumina_ffn_width(384),
                      predict_size=32, sample_num=32, num_inference_steps=10, memory_frames=2)
"""DualVLN System-1 (`nextdit_async`): NextDiTCrossAttnConfig defaults (nextdit_crossattn_traj.py:12-30) with
latent_embedding_size 768 (internvla_n1_arch.py:6,22), generate_traj defaults (internvla_n1.py:354-357). FFN width 1536 = the reference's
`inner_dim=4 * dim` under its pinned diffusers==0.33.1; UNVERIFIED until a 0.33.1 wheel or the released checkpoint's key list is seen."""

N1_NEXTDIT_CFG_FFN1024 = dict(N1_NEXTDIT_CFG, dit_ffn=lumina_ffn_width(384, legacy_two_thirds=True))
"""the same head under the diffusers <= 0.32 convention (FFN 1024): what rounds 1-5 built and measured; kept pinned and benchmarkable."""

N1_NEXTDIT_VARIANTS = {"ffn1536": N1_NEXTDIT_CFG, "ffn1024": N1_NEXTDIT_CFG_FFN1024}


def n1_nextdit_cfg_from_weights(weights, prefix: str = "", base=None) -> dict:
    """System-1 geometry read off the checkpoint's own tensor shapes (nothing about the DiT is in config.json: NextDiTCrossAttnConfig is
    constructed in code, internvla_n1_arch.py:127-131). `weights`: mapping name -> tensor-like with `.shape` (or a `shape_of(name)` method)."""
    def shape(k):
        k = prefix + k
        if hasattr(weights, "shape_of"):
            return tuple(weights.shape_of(k))
        return tuple(weights[k].shape)

    def has(k):
        return (prefix + k) in weights

    cfg = dict(base or N1_NEXTDIT_CFG)
    p = "traj_dit.model.layers."
    ffn, D = shape(p + "0.feed_forward.linear_1.weight")
    nl = 0
    while has(f"{p}{nl}.attn1.to_q.weight"):
        nl += 1
    heads = shape(p + "0.gate")[0]
    L, V = shape("cond_projector.0.weight")
    assert shape(p + "0.attn1.to_q.weight") == (D, D) and shape(p + "0.feed_forward.linear_2.weight") == (D, ffn), "inconsistent NextDiT tensor shapes"
    assert shape("traj_dit.model.caption_projection.linear_1.weight") == (D, L) and D % heads == 0
    cfg.update(dit_dim=D, dit_ffn=ffn, dit_layers=nl, dit_heads=heads, latent_dim=L, vlm_token_dim=V)
    return cfg


def n1_nextdit_spec(cfg=N1_NEXTDIT_CFG) -> Spec:
    """System-1 parameters of InternVLAN1Model for system1 = 'nextdit_async' (internvla_n1_arch.py:127-141)."""
    V, L, D = cfg["vlm_token_dim"], cfg["latent_dim"], cfg["dit_dim"]
    s: Spec = {}
    _lin(s, "cond_projector.0", L, V)
    _lin(s, "cond_projector.2", L, L)
    s.update(dinov2_vits_spec("rgb_model."))
    s["memory_encoder.memory_pos"] = ((512, D), "emb")
    for i in range(3):
        s.update(encoder_layer_spec(f"memory_encoder.encoder.layers.{i}", D, 2048))
    s["rgb_resampler.query_tokens"] = ((32, L), "latent")
    s["rgb_resampler.query_pos"] = ((32, L), "emb")
    for i in range(3):
        s.update(decoder_layer_spec(f"rgb_resampler.decoder.layers.{i}", L, 2048))
    _lin(s, "action_encoder", D, 3)
    _lin(s, "action_decoder", 3, D, kind="w_small")
    p = "traj_dit.model."
    _lin(s, p + "caption_projection.linear_1", D, L)
    _lin(s, p + "caption_projection.linear_2", D, D)
    _lin(s, p + "time_caption_embed.timestep_embedder.linear_1", D, 256)
    _lin(s, p + "time_caption_embed.timestep_embedder.linear_2", D, D)
    _ln(s, p + "time_caption_embed.caption_embedder.0", D)
    _lin(s, p + "time_caption_embed.caption_embedder.1", D, D)
    for i in range(cfg["dit_layers"]):
        b = f"{p}layers.{i}"
        s[b + ".gate"] = ((cfg["dit_heads"],), "gate")
        for a in ("attn1", "attn2"):
            _ln(s, f"{b}.{a}.norm_q", D)
            _ln(s, f"{b}.{a}.norm_k", D)
            for n in ("to_q", "to_k", "to_v"):
                _lin(s, f"{b}.{a}.{n}", D, D, bias=False)
        _lin(s, b + ".attn2.to_out.0", D, D, bias=False)
        _lin(s, b + ".feed_forward.linear_1", cfg["dit_ffn"], D, bias=False)
        _lin(s, b + ".feed_forward.linear_2", D, cfg["dit_ffn"], bias=False)
        _lin(s, b + ".feed_forward.linear_3", cfg["dit_ffn"], D, bias=False)
        s[b + ".norm1.linear.weight"] = ((4 * D, D), "w_small")
        s[b + ".norm1.linear.bias"] = ((4 * D,), "b")
        for n in ("norm1.norm", "ffn_norm1", "norm2", "ffn_norm2", "norm1_context"):
            _ln(s, f"{b}.{n}", D, bias=False)
    s[p + "norm_out.linear_1.weight"] = ((D, D), "w_small")
    s[p + "norm_out.linear_1.bias"] = ((D,), "b")
    _lin(s, p + "norm_out.linear_2", D, D)
    return s


UNET1D_CFG = dict(input_dim=3, global_cond_dim=384, dsed=256, down_dims=(256, 512, 1024), kernel_size=5, n_groups=8, predict_size=32,
                  sample_num=32, num_train_timesteps=100, num_inference_steps=10)
"""diffusion-policy ConditionalUnet1D as an alternative System-1 head (SURVEY.md 8f-3): network hyper-parameters and DDIM schedule of the
vendored config (diffusion_policy/config/train_diffusion_unet_ddim_lowdim_workspace.yaml:32-49: down_dims [256,512,1024], kernel 5,
8 groups, cond_predict_scale, DDIM over 100 train steps), 3-d waypoints x 32 steps x 32 samples like the N1 heads, 10 DDIM steps
(BASELINE.json north_star), conditioned on one 384-d vector per env."""


def unet1d_spec(cfg=UNET1D_CFG) -> Spec:
    """ConditionalUnet1D parameters (conditional_unet1d.py:69-187), cond_predict_scale=True, no local conditioning."""
    s: Spec = {}
    k, dsed = cfg["kernel_size"], cfg["dsed"]
    cond = dsed + cfg["global_cond_dim"]
    dims = [cfg["input_dim"]] + list(cfg["down_dims"])

    def conv(p, co, ci, kk):
        s[p + ".weight"] = ((co, ci, kk), "w")
        s[p + ".bias"] = ((co,), "b")

    def res(p, ci, co):
        for j, c_in in ((0, ci), (1, co)):
            conv(f"{p}.blocks.{j}.block.0", co, c_in, k)
            _ln(s, f"{p}.blocks.{j}.block.1", co)
        _lin(s, p + ".cond_encoder.1", 2 * co, cond, kind="w_small")
        if ci != co:
            conv(p + ".residual_conv", co, ci, 1)

    _lin(s, "diffusion_step_encoder.1", 4 * dsed, dsed)
    _lin(s, "diffusion_step_encoder.3", dsed, 4 * dsed)
    n = len(dims) - 1
    for i in range(n):
        res(f"down_modules.{i}.0", dims[i], dims[i + 1])
        res(f"down_modules.{i}.1", dims[i + 1], dims[i + 1])
        if i < n - 1:
            conv(f"down_modules.{i}.2.conv", dims[i + 1], dims[i + 1], 3)
    for i in range(2):
        res(f"mid_modules.{i}", dims[-1], dims[-1])
    for i, (di, do) in enumerate(reversed(list(zip(dims[1:-1], dims[2:])))):
        res(f"up_modules.{i}.0", 2 * do, di)
        res(f"up_modules.{i}.1", di, di)
        s[f"up_modules.{i}.2.conv.weight"] = ((di, di, 4), "w")      # ConvTranspose1d weight [in, out, k]
        s[f"up_modules.{i}.2.conv.bias"] = ((di,), "b")
    conv("final_conv.0.block.0", dims[1], dims[1], k)
    _ln(s, "final_conv.0.block.1", dims[1])
    conv("final_conv.1", cfg["input_dim"], dims[1], 1)
    return s


def unet1d_inputs(B: int, seed: int = 0, cfg=UNET1D_CFG):
    g = torch.Generator().manual_seed(5000 + seed)
    return dict(global_cond=torch.randn(B, cfg["global_cond_dim"], generator=g).to(torch.bfloat16).float(),
                x_init=torch.randn(B, cfg["sample_num"], cfg["predict_size"], cfg["input_dim"], generator=g))


def n1_full_spec(qwen_cfg=None, system1: str = "nextdit_async", s1_cfg=None) -> Spec:
    """every parameter of an InternVLA-N1 checkpoint: Qwen2.5-VL (visual.*, model.*, lm_head) + the System-1 modules under `model.`."""
    s = qwen_spec(qwen_cfg or QWEN_N1_CFG)
    s1 = n1_nextdit_spec(s1_cfg or N1_NEXTDIT_CFG) if "nextdit" in system1 else {("navdp." + k): v for k, v in n1_navdp_spec().items()}
    if "async" not in system1:
        # the plain 'nextdit' / 'navdp' types condition on the VLM latents alone (internvla_n1.py:382-383, navdp.py:255-289): the look-down
        # memory modules are not part of the engines' needs
        drop = ("rgb_model.", "memory_encoder.", "rgb_resampler.") if "nextdit" in system1 else ("navdp.rgbd_encoder.", "navdp.goal_compressor.")
        s1 = {k: v for k, v in s1.items() if not k.startswith(drop)}
    s.update({"model." + k: v for k, v in s1.items()})
    return s


def n1_nextdit_state_dict(seed: int = 0, cfg=N1_NEXTDIT_CFG):
    return materialize(n1_nextdit_spec(cfg), seed)


def n1_nextdit_inputs(B: int, seed: int = 0, cfg=N1_NEXTDIT_CFG):
    """traj_latents as produced by generate_latents (bf16 hidden states), 2 look-down frames in 0..1 (bf16 in the reference), init noise."""
    g = torch.Generator().manual_seed(3000 + seed)
    lat = torch.randn(B, cfg["n_query"], cfg["vlm_token_dim"], generator=g).to(torch.bfloat16).float()
    images = torch.rand(B, cfg["memory_frames"], 224, 224, 3, generator=g).to(torch.bfloat16).float()
    x_init = torch.randn(B, cfg["sample_num"], cfg["predict_size"], 3, generator=g)
    return dict(traj_latents=lat, images=images, x_init=x_init)


QWEN_N1_CFG = dict(v_hidden=1280, v_heads=16, v_inter=3420, v_depth=32, v_fullatt=(7, 15, 23, 31), v_window=112, v_patch=14,
                   v_out=3584, t_hidden=3584, t_inter=18944, t_heads=28, t_kv_heads=4, t_layers=28, vocab=152064,
                   rope_theta=1e6, n_query=4, image_token_id=151655, traj_token_id=151667, vision_start_id=151652,
                   vision_end_id=151653, eos_token_id=151645)
"""InternVLA-N1 System-2 = Qwen2.5-VL-7B dims (hidden 3584 hard-coded at internvla_n1_arch.py:133, navdp.py:25) with the token ids
the reference hard-codes (internvla_n1.py:18-19)."""

QWEN_TEST_CFG = dict(QWEN_N1_CFG, v_depth=2, v_fullatt=(1,), t_layers=2, vocab=4096, image_token_id=4001, traj_token_id=4002,
                     vision_start_id=4003, vision_end_id=4004, eos_token_id=4005)
"""Parity-test configuration: true widths / head geometry of every layer type (window + full ViT block, GQA decoder layer), reduced
depth and vocabulary so the fp32 CPU oracle and the transformers modules that pin it run in seconds."""


def qwen_spec(cfg) -> Spec:
    """Qwen2.5-VL parameters under the reference checkpoint's names (transformers 4.51 layout) + model.latent_queries."""
    s: Spec = {}
    D, I, O = cfg["v_hidden"], cfg["v_inter"], cfg["v_out"]
    s["visual.patch_embed.proj.weight"] = ((D, 3, 2, 14, 14), "w")
    for i in range(cfg["v_depth"]):
        b = f"visual.blocks.{i}."
        _ln(s, b + "norm1", D, bias=False)
        _ln(s, b + "norm2", D, bias=False)
        _lin(s, b + "attn.qkv", 3 * D, D)
        _lin(s, b + "attn.proj", D, D)
        _lin(s, b + "mlp.gate_proj", I, D)
        _lin(s, b + "mlp.up_proj", I, D)
        _lin(s, b + "mlp.down_proj", D, I)
    _ln(s, "visual.merger.ln_q", D, bias=False)
    _lin(s, "visual.merger.mlp.0", 4 * D, 4 * D)
    _lin(s, "visual.merger.mlp.2", O, 4 * D)
    H, TI, nh, nkv = cfg["t_hidden"], cfg["t_inter"], cfg["t_heads"], cfg["t_kv_heads"]
    hd = H // nh
    s["model.embed_tokens.weight"] = ((cfg["vocab"], H), "latent")
    for i in range(cfg["t_layers"]):
        b = f"model.layers.{i}."
        _ln(s, b + "input_layernorm", H, bias=False)
        _ln(s, b + "post_attention_layernorm", H, bias=False)
        _lin(s, b + "self_attn.q_proj", nh * hd, H)
        _lin(s, b + "self_attn.k_proj", nkv * hd, H)
        _lin(s, b + "self_attn.v_proj", nkv * hd, H)
        _lin(s, b + "self_attn.o_proj", H, nh * hd, bias=False)
        _lin(s, b + "mlp.gate_proj", TI, H, bias=False)
        _lin(s, b + "mlp.up_proj", TI, H, bias=False)
        _lin(s, b + "mlp.down_proj", H, TI, bias=False)
    _ln(s, "model.norm", H, bias=False)
    s["model.latent_queries"] = ((1, cfg["n_query"], H), "latent")
    _lin(s, "lm_head", cfg["vocab"], H, bias=False)
    return s


def qwen_state_dict(seed: int = 0, cfg=QWEN_TEST_CFG):
    return materialize(qwen_spec(cfg), seed)


def qwen_inputs(B: int, n_img: int, seed: int = 0, cfg=QWEN_TEST_CFG, grid=(1, 28, 28), n_text: int = 24, n_tail: int = 8):
    """Synthetic S2 prompt per env: [text | (<vision_start> <image>*g <vision_end>) * n_img | tail text] with
    pixel_values ~ N(0,1) [B*n_img*h*w, 1176] already in the HF processor's patch layout (a1 stays on the host)."""
    g = torch.Generator().manual_seed(4000 + seed)
    t, h, w = grid
    per = t * h * w // 4
    lim = min(cfg["vocab"], cfg["image_token_id"]) - 8
    rows = []
    for _ in range(B):
        ids = torch.randint(0, lim, (n_text,), generator=g).tolist()
        for _ in range(n_img):
            ids += [cfg["vision_start_id"]] + [cfg["image_token_id"]] * per + [cfg["vision_end_id"]]
        ids += torch.randint(0, lim, (n_tail,), generator=g).tolist()
        rows.append(ids)
    input_ids = torch.tensor(rows, dtype=torch.long)
    pixel_values = torch.randn(B * n_img * t * h * w, 1176, generator=g).to(torch.bfloat16).float()
    grid_thw = torch.tensor([list(grid)] * (B * n_img), dtype=torch.long)
    return dict(input_ids=input_ids, pixel_values=pixel_values, grid_thw=grid_thw)


def qwen_lookdown_inputs(cfg=QWEN_TEST_CFG):
    """S2 prompt with the un-resized look-down frame of the reference's callers (internvla_n1_policy.py:113-116,140):
    text | <vs> 196 img <ve> | text | <vs> 391 img <ve> | tail, grids (1,28,28) and (1,34,46) (476x644 pixels -> ragged windows)."""
    g = torch.Generator().manual_seed(4242)
    lim = cfg["image_token_id"] - 8
    ids = torch.randint(0, lim, (12,), generator=g).tolist()
    for n in (196, 391):
        ids += [cfg["vision_start_id"]] + [cfg["image_token_id"]] * n + [cfg["vision_end_id"]] + torch.randint(0, lim, (5,), generator=g).tolist()
    grid = torch.tensor([[1, 28, 28], [1, 34, 46]])
    pv = torch.randn(28 * 28 + 34 * 46, 1176, generator=g).to(torch.bfloat16).float()
    return dict(input_ids=torch.tensor([ids]), pixel_values=pv, grid_thw=grid)


def navdpnet_state_dict(seed: int = 0, cfg=NAVDPNET_CFG):
    return materialize(navdpnet_spec(cfg), seed)


def n1_navdp_state_dict(seed: int = 0, cfg=N1_NAVDP_CFG):
    return materialize(n1_navdp_spec(cfg), seed)


# ------------------------------------------------------------------------------------------------------ synthetic inputs
def navdpnet_inputs(B: int, seed: int = 0, cfg=NAVDPNET_CFG):
    """Synthetic config-#2 inputs: point goal, memory_size RGB frames in 0..1, one depth frame in metres
    (clipped to 5 m), initial noise and the per-step DDPM noise, all from a CPU generator."""
    g = torch.Generator().manual_seed(1000 + seed)
    M, T, S, K = cfg["memory_size"], cfg["predict_size"], cfg["sample_num"], cfg["num_train_timesteps"]
    goal = torch.randn(B, 3, generator=g) * torch.tensor([3.0, 3.0, 0.5])
    images = torch.rand(B, M, 224, 224, 3, generator=g)
    depths = torch.rand(B, 1, 224, 224, 1, generator=g) * 5.0
    x_init = torch.randn(B, S, T, 3, generator=g)
    step_noise = torch.randn(K, B, S, T, 3, generator=g)
    return dict(goal=goal, images=images, depths=depths, x_init=x_init, step_noise=step_noise)


def n1_navdp_inputs(B: int, seed: int = 0, cfg=N1_NAVDP_CFG):
    g = torch.Generator().manual_seed(2000 + seed)
    M, T, S, K = cfg["memory_size"], cfg["predict_size"], cfg["sample_num"], cfg["num_train_timesteps"]
    vlm = torch.randn(B, 4, cfg["vlm_token_dim"], generator=g).to(torch.bfloat16).float()
    images = torch.rand(B, M, 224, 224, 3, generator=g)
    depths = torch.rand(B, M, 224, 224, 1, generator=g) * 5.0
    x_init = torch.randn(B, S, T, 3, generator=g)
    step_noise = torch.randn(K, B, S, T, 3, generator=g)
    return dict(vlm_tokens=vlm, images=images, depths=depths, x_init=x_init, step_noise=step_noise)


def write_checkpoint(path, qwen_cfg=None, system1: str = "nextdit_async", seed: int = 0, shards: int = 2, s1_cfg=None):
    """A synthetic InternVLA-N1 checkpoint ON DISK in the layout of a real one (HF safetensors shards with the reference's parameter
    names + config.json in the Qwen2.5-VL / InternVLAN1ModelConfig layout): what `InternVLAN1ForCausalLM.from_pretrained` and the
    agent's config-only construction are tested against (no real checkpoint is available offline). bf16 tensors, like the release."""
    import json
    from pathlib import Path

    from safetensors.torch import save_file

    cfg = qwen_cfg or QWEN_TEST_CFG
    p = Path(path)
    p.mkdir(parents=True, exist_ok=True)
    # (s1_cfg: the NextDiT geometry the tensors are written at - it is NOT recorded in config.json, as in the reference's checkpoints)
    sd = {k: v.to(torch.bfloat16).contiguous() for k, v in materialize(n1_full_spec(cfg, system1, s1_cfg=s1_cfg), seed).items()}
    keys = sorted(sd)
    per = (len(keys) + shards - 1) // shards
    for i in range(shards):
        save_file({k: sd[k] for k in keys[i * per:(i + 1) * per]}, str(p / f"model-{i + 1:05d}-of-{shards:05d}.safetensors"))
    hf = {"architectures": ["InternVLAN1ForCausalLM"], "model_type": "internvla_n1", "system1": system1, "n_query": cfg["n_query"],
          "hidden_size": cfg["t_hidden"], "intermediate_size": cfg["t_inter"], "num_hidden_layers": cfg["t_layers"],
          "num_attention_heads": cfg["t_heads"], "num_key_value_heads": cfg["t_kv_heads"], "vocab_size": cfg["vocab"],
          "rope_theta": cfg["rope_theta"], "rms_norm_eps": 1e-6, "eos_token_id": cfg["eos_token_id"], "image_token_id": cfg["image_token_id"],
          "traj_token_id": cfg["traj_token_id"], "vision_start_token_id": cfg["vision_start_id"], "vision_end_token_id": cfg["vision_end_id"],
          "rope_scaling": {"type": "mrope", "mrope_section": [16, 24, 24]}, "torch_dtype": "bfloat16",
          "vision_config": {"depth": cfg["v_depth"], "hidden_size": cfg["v_hidden"], "intermediate_size": cfg["v_inter"], "num_heads": cfg["v_heads"],
                            "out_hidden_size": cfg["v_out"], "fullatt_block_indexes": list(cfg["v_fullatt"]), "window_size": cfg["v_window"],
                            "patch_size": cfg["v_patch"], "spatial_merge_size": 2}}
    (p / "config.json").write_text(json.dumps(hf, indent=1))
    return sd
