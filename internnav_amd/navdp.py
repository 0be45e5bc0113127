"""NavDP System-1 policies (diffusion trajectory heads) on the gfx950 op library - batched over environments.

Two engines with the reference's method names and tensor contracts:
  * `NavDPNet`                        <- internnav/model/basemodel/navdp/navdp_policy.py:34  (BASELINE config #2)
        predict_pointgoal_batch_action_vel(goal_point, input_images, input_depths) -> (negative, positive) trajectories
  * `NavDPPolicyDAT` (N1 navdp_async) <- internnav/model/basemodel/internvla_n1/navdp.py:16
        predict_pointgoal_action_async(vlm_tokens, input_images, input_depths) -> all 32 sampled trajectories

The reference executes one environment per call (navdp_policy.py:165 / navdp.py:228-231); here every call carries B
environments x 32 samples. Semantics per environment are identical (oracle = loop of batch-1 calls).
Step-invariant work is hoisted out of the denoising loop without changing results:
  * the RGB-D tokens, the goal embedding and cond rows 1.. are computed once; only the time row (row 0) of the condition
    changes per step (navdp_policy.py:161-164), so it is refreshed from a precomputed table [steps, C];
  * the 32 samples of one environment share the condition: cross-attention K/V are computed once per env (B x L rows) and
    broadcast to the 32 sample sequences by the attention kernel (kv_bdiv = 32) instead of `cond.repeat(32*B, 1, 1)`.
Sampler noise is an explicit input (the reference draws torch.randn inside the loop) so results are reproducible and
checkable against the CPU oracle.

HBM layout (B envs, S = 32 samples, T = predict_size, L = condition length, C = 384):
  cond    bf16 [B*L, C]        condition tokens per env: [time | goal x g | rgbd tokens] + positional table
  kv[l]   bf16 [B*L, 2C]       per-layer cross-attention K|V of the condition
  x       f32  [B*S*T, C]      residual stream of the denoiser, updated in place by the GEMM epilogues
  h / att bf16 [B*S*T, C] , qkv bf16 [B*S*T, 3C] , ff bf16 [B*S*T, 4C]
  sample  f32  [B*S*T, 3]      the trajectories being denoised (updated in place by the fused head + scheduler kernel)
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

from . import ops
from .vit_s import DinoV2Encoder, VitWorkspace

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
# DAT_RGBD_Patch_Backbone holds its constants in bf16 (input_dtype="bf16" default, navdp_backbone.py:119-127)
# (round 4: the two tuples were typed in by hand and had TWO wrong entries - mean G 0.45703125, std R 0.2294921875 - which alone put the
# N1 NavDP head further from fp32 than bf16-autocast PyTorch; they are now derived from the fp32 constants by the same cast the reference does)
IMAGENET_MEAN_BF16 = tuple(float(torch.tensor(v, dtype=torch.float32).to(torch.bfloat16)) for v in IMAGENET_MEAN)   # (0.484375, 0.455078125, 0.40625)
IMAGENET_STD_BF16 = tuple(float(torch.tensor(v, dtype=torch.float32).to(torch.bfloat16)) for v in IMAGENET_STD)     # (0.228515625, 0.2236328125, 0.224609375)


def ddpm_tables(num_train_timesteps: int):
    """DDPMScheduler(squaredcos_cap_v2, epsilon, clip_sample, fixed_small, leading) with num_inference_steps ==
    num_train_timesteps (navdp_policy.py:119-121,311; navdp.py:74-76,246): per step (descending t) the five scalars of
    x0 = clamp((x - sqrt(1-abar) eps) / sqrt(abar)); x' = c_x0 x0 + c_xt x + sigma z. Pure host scalar math (float32 like
    diffusers), fed to the fused head kernel."""
    K = num_train_timesteps

    def alpha_bar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2

    betas = torch.tensor([min(1 - alpha_bar((i + 1) / K) / alpha_bar(i / K), 0.999) for i in range(K)], dtype=torch.float32)
    acp = torch.cumprod(1.0 - betas, dim=0)
    one = torch.tensor(1.0)
    steps = []
    for t in range(K - 1, -1, -1):
        a_t = acp[t]
        a_prev = acp[t - 1] if t - 1 >= 0 else one
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20)
        steps.append((t, (float(1.0 / a_t.sqrt()), float(b_t.sqrt()), float(a_prev.sqrt() * cur_b / b_t),
                          float(cur_a.sqrt() * b_prev / b_t), float(var.sqrt()) if t > 0 else 0.0)))
    return steps


def sinusoidal_pos_emb(t: float, dim: int) -> torch.Tensor:
    """SinusoidalPosEmb(dim)(t) (navdp_backbone.py:9-21): input-independent, tabulated on the host per timestep."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half, dtype=torch.float32) * -e) * float(t)
    return torch.cat((f.sin(), f.cos()))


class _DecoderLayer:
    """Weights of one nn.TransformerDecoderLayer (packed in_proj split into the GEMMs actually issued)."""

    def __init__(self, sd, p, device, d):
        bf, f32 = torch.bfloat16, torch.float32

        def w(t):
            return t.to(device=device, dtype=bf).contiguous()

        def f(t):
            return t.to(device=device, dtype=f32).contiguous()

        self.sa_w, self.sa_b = w(sd[p + ".self_attn.in_proj_weight"]), f(sd[p + ".self_attn.in_proj_bias"])
        self.sa_ow, self.sa_ob = w(sd[p + ".self_attn.out_proj.weight"]), f(sd[p + ".self_attn.out_proj.bias"])
        W, b = sd[p + ".multihead_attn.in_proj_weight"], sd[p + ".multihead_attn.in_proj_bias"]
        self.ca_qw, self.ca_qb = w(W[:d]), f(b[:d])
        self.ca_kvw, self.ca_kvb = w(W[d:]), f(b[d:])
        self.ca_ow, self.ca_ob = w(sd[p + ".multihead_attn.out_proj.weight"]), f(sd[p + ".multihead_attn.out_proj.bias"])
        self.l1w, self.l1b = w(sd[p + ".linear1.weight"]), f(sd[p + ".linear1.bias"])
        self.l2w, self.l2b = w(sd[p + ".linear2.weight"]), f(sd[p + ".linear2.bias"])
        self.n = [(f(sd[f"{p}.norm{i}.weight"]), f(sd[f"{p}.norm{i}.bias"])) for i in (1, 2, 3)]


class _SeqWorkspace:
    def __init__(self, rows: int, d: int, ffn: int, device):
        bf = torch.bfloat16
        self.x = torch.empty(rows, d, dtype=torch.float32, device=device)
        self.h = torch.empty(rows, d, dtype=bf, device=device)
        self.att = torch.empty(rows, d, dtype=bf, device=device)
        self.qkv = torch.empty(rows, 3 * d, dtype=bf, device=device)
        self.ff = torch.empty(rows, ffn, dtype=bf, device=device)


def _attn_views(buf, nseq, L, nh, hd, ncomp):
    return buf.view(nseq, L, ncomp, nh, hd)


def decoder_layer_prenorm(L: _DecoderLayer, ws: _SeqWorkspace, nseq: int, T: int, kv: torch.Tensor, n_mem: int, Lm: int,
                          nh: int, act: str, causal: bool, kv_start: int = 0, eps: float = 1e-5):
    """nn.TransformerDecoderLayer(norm_first=True) forward on the f32 residual ws.x [nseq*T, d]; `kv` = this layer's
    cross-attention K|V of the memory, bf16 [n_mem*Lm, 2d]; sequences i*(nseq/n_mem) .. share memory i."""
    rows = nseq * T
    d = ws.x.shape[1]
    hd = d // nh
    x, h, att, qkv, ff = ws.x[:rows], ws.h[:rows], ws.att[:rows], ws.qkv[:rows], ws.ff[:rows]
    q5 = qkv.view(nseq, T, 3, nh, hd)
    kv5 = kv.view(n_mem, Lm, 2, nh, hd)
    a4 = att.view(nseq, T, nh, hd)
    ops.norm(x, L.n[0][0], L.n[0][1], eps=eps, out=h)
    ops.linear(h, L.sa_w, bias=L.sa_b, out=qkv)
    ops.attention(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], causal=causal, out=a4)
    ops.linear(att, L.sa_ow, bias=L.sa_ob, residual=x, out=x)
    ops.norm(x, L.n[1][0], L.n[1][1], eps=eps, out=h)
    qc = qkv[:, :d]
    ops.linear(h, L.ca_qw, bias=L.ca_qb, out=qc)
    # the nseq / n_mem sequences that share one memory are contiguous rows: present them as ONE query sequence per memory, so a
    # workgroup stages the memory's K/V once per 64 query rows instead of once per (short) sequence
    per = (nseq // n_mem) * T
    ops.attention(qc.view(n_mem, per, nh, hd), kv5[:, :, 0], kv5[:, :, 1], kv_start=kv_start, out=att.view(n_mem, per, nh, hd))
    ops.linear(att, L.ca_ow, bias=L.ca_ob, residual=x, out=x)
    ops.norm(x, L.n[2][0], L.n[2][1], eps=eps, out=h)
    ops.linear(h, L.l1w, bias=L.l1b, act=act, out=ff)
    ops.linear(ff, L.l2w, bias=L.l2b, residual=x, out=x)


def decoder_layer_postnorm(L: _DecoderLayer, ws: _SeqWorkspace, nseq: int, T: int, kv: torch.Tensor, n_mem: int, Lm: int,
                           nh: int, act: str, eps: float = 1e-5):
    """nn.TransformerDecoderLayer(norm_first=False): x = LN(x + sublayer(x)). ws.x holds the f32 stream, ws.h its bf16 copy."""
    rows = nseq * T
    d = ws.x.shape[1]
    hd = d // nh
    x, h, att, qkv, ff = ws.x[:rows], ws.h[:rows], ws.att[:rows], ws.qkv[:rows], ws.ff[:rows]
    q5 = qkv.view(nseq, T, 3, nh, hd)
    kv5 = kv.view(n_mem, Lm, 2, nh, hd)
    a4 = att.view(nseq, T, nh, hd)
    ops.linear(h, L.sa_w, bias=L.sa_b, out=qkv)
    ops.attention(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], out=a4)
    ops.linear(att, L.sa_ow, bias=L.sa_ob, residual=x, out=x)
    ops.norm(x, L.n[0][0], L.n[0][1], eps=eps, out=h, out32=x)
    qc = qkv[:, :d]
    ops.linear(h, L.ca_qw, bias=L.ca_qb, out=qc)
    ops.attention(qc.view(nseq, T, nh, hd), kv5[:, :, 0], kv5[:, :, 1], kv_bdiv=nseq // n_mem, out=a4)
    ops.linear(att, L.ca_ow, bias=L.ca_ob, residual=x, out=x)
    ops.norm(x, L.n[1][0], L.n[1][1], eps=eps, out=h, out32=x)
    ops.linear(h, L.l1w, bias=L.l1b, act=act, out=ff)
    ops.linear(ff, L.l2w, bias=L.l2b, residual=x, out=x)
    ops.norm(x, L.n[2][0], L.n[2][1], eps=eps, out=h, out32=x)


class _RGBDFormer:
    """former_net (2 post-LN decoder layers, 8 heads, ReLU FFN 2048) + project_layer of the RGB-D backbones
    (navdp_backbone.py:148-149,199-202 / :243-246,281-285)."""

    def __init__(self, sd, p, device, n_query: int, n_tokens: int, b_max: int, query_key: str, pe_key: str):
        self.layers = [_DecoderLayer(sd, f"{p}former_net.layers.{i}", device, 384) for i in range(2)]
        self.proj_w = sd[p + "project_layer.weight"].to(device=device, dtype=torch.bfloat16).contiguous()
        self.proj_b = sd[p + "project_layer.bias"].to(device=device, dtype=torch.float32).contiguous()
        self.query = sd[p + query_key].to(device=device, dtype=torch.float32).contiguous()          # [n_query, 384]
        self.pe = sd[p + pe_key][:n_tokens].to(device=device, dtype=torch.float32).contiguous()     # [n_tokens, 384]
        self.nq, self.nt = n_query, n_tokens
        self.tokens = torch.empty(b_max * n_tokens, 384, dtype=torch.bfloat16, device=device)
        self.kv = torch.empty(b_max * n_tokens, 768, dtype=torch.bfloat16, device=device)
        self.ws = _SeqWorkspace(b_max * n_query, 384, 2048, device)

    def forward(self, B: int, out3: torch.Tensor, residual: Optional[torch.Tensor]):
        """tokens (already filled: ViT tokens + former_pe) -> out3 [B, n_query, token_dim] view (+ residual table)."""
        ws, nq, nt = self.ws, self.nq, self.nt
        rows = B * nq
        ops.embed3(None, None, None, out=ws.x[:rows], pos=self.query, rows=rows)
        ops.embed3(None, None, None, out=ws.h[:rows], pos=self.query, rows=rows)
        tok = self.tokens[: B * nt]
        for L in self.layers:
            ops.linear(tok, L.ca_kvw, bias=L.ca_kvb, out=self.kv[: B * nt])
            decoder_layer_postnorm(L, ws, B, nq, self.kv[: B * nt], B, nt, 8, "relu")
        ops.linear(ws.h[:rows].view(B, nq, 384), self.proj_w, bias=self.proj_b, residual=residual, out=out3, batched=True)


class _NavDPBase:
    """Shared denoiser machinery: 16 pre-LN decoder layers over B*S sequences of T action tokens."""

    def _init_denoiser(self, sd, device, cfg, b_max: int, cond_len: int):
        D, T, S = cfg["token_dim"], cfg["predict_size"], cfg["sample_num"]
        self.D, self.T, self.S, self.Lc, self.b_max, self.device = D, T, S, cond_len, b_max, device
        self.heads, self.depth = cfg["heads"], cfg["temporal_depth"]
        self.layers = [_DecoderLayer(sd, f"decoder.layers.{i}", device, D) for i in range(self.depth)]
        f32 = torch.float32
        self.in_w = sd["input_embed.weight"].to(device=device, dtype=f32).contiguous()
        self.in_b = sd["input_embed.bias"].to(device=device, dtype=f32).contiguous()
        self.ln_w = sd["layernorm.weight"].to(device=device, dtype=f32).contiguous()
        self.ln_b = sd["layernorm.bias"].to(device=device, dtype=f32).contiguous()
        self.head_w = sd["action_head.weight"].to(device=device, dtype=f32).contiguous()
        self.head_b = sd["action_head.bias"].to(device=device, dtype=f32).contiguous()
        self.steps = ddpm_tables(cfg["num_train_timesteps"])
        self.cond = torch.empty(b_max * cond_len, D, dtype=torch.bfloat16, device=device)
        self.kv = [torch.empty(b_max * cond_len, 2 * D, dtype=torch.bfloat16, device=device) for _ in range(self.depth)]
        self.ws = _SeqWorkspace(b_max * S * T, D, 4 * D, device)
        self.sample = torch.empty(b_max * S * T, 3, dtype=f32, device=device)

    def _set_time_tables(self, cond_pos0: torch.Tensor):
        """cond row 0 = SinusoidalPosEmb(t) + cond_pos_embed[0] for every sampler step, tabulated once (input independent)."""
        rows = [sinusoidal_pos_emb(t, self.D) + cond_pos0.float().cpu() for t, _ in self.steps]
        self.time_rows = torch.stack(rows).to(self.device).contiguous()  # [K, D] f32

    def _denoise(self, B: int, x_init: torch.Tensor, step_noise: torch.Tensor):
        S, T, D, Lc = self.S, self.T, self.D, self.Lc
        rows = B * S * T
        sample = self.sample[: rows]
        sample.copy_(x_init.reshape(rows, 3))
        cond = self.cond[: B * Lc]
        for i, (t, coef) in enumerate(self.steps):
            ops.embed3(None, None, None, out=cond, pos=self.time_rows[i:i + 1], rows=B, out_map=(1, Lc, 0))
            ops.embed3(sample, self.in_w, self.in_b, out=self.ws.x[:rows], pos=self.out_pos)
            for l, L in enumerate(self.layers):
                ops.linear(cond, L.ca_kvw, bias=L.ca_kvb, out=self.kv[l][: B * Lc])
                decoder_layer_prenorm(L, self.ws, B * S, T, self.kv[l][: B * Lc], B, Lc, self.heads, "gelu", causal=True)
            noise = step_noise[i].reshape(rows, 3) if t > 0 else None
            ops.head3(self.ws.x[:rows], self.head_w, self.head_b, self.ln_w, self.ln_b, eps=1e-5, mode=1, sample=sample,
                      noise=noise, coef=coef, clip=1.0)
        return sample


class NavDPNet(_NavDPBase):
    """MI355X engine behind `NavDPNet.predict_pointgoal_batch_action_vel` (navdp_policy.py:302-321), batched over envs."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: dict, device="cuda:0", max_envs: int = 64):
        device = torch.device(device)
        self.cfg = cfg
        M = cfg["memory_size"]
        self.M = M
        Lc = M * 16 + 4
        self._init_denoiser(state_dict, device, cfg, max_envs, Lc)
        sd, p = state_dict, "rgbd_encoder."
        self.rgb = DinoV2Encoder(sd, p + "rgb_model.", device)
        self.depth_vit = DinoV2Encoder(sd, p + "depth_model.", device)
        self.vit_ws = VitWorkspace(max_envs * M, device)
        self.former = _RGBDFormer(sd, p, device, M * 16, (M + 1) * 256, max_envs, "former_query.position_embedding.weight",
                                  "former_pe.position_embedding.weight")
        f32 = torch.float32
        self.cond_pos = sd["cond_pos_embed.position_embedding.weight"].to(device=device, dtype=f32).contiguous()  # [Lc, D]
        self.out_pos = sd["out_pos_embed.position_embedding.weight"].to(device=device, dtype=f32).contiguous()    # [T, D]
        self.pt_w = sd["point_encoder.weight"].to(device=device, dtype=f32).contiguous()
        self.pt_b = sd["point_encoder.bias"].to(device=device, dtype=f32).contiguous()
        self.cr_w = sd["critic_head.weight"].to(device=device, dtype=f32).contiguous()
        self.cr_b = sd["critic_head.bias"].to(device=device, dtype=f32).contiguous()
        self._set_time_tables(sd["cond_pos_embed.position_embedding.weight"][0])
        self.critic = torch.empty(max_envs * self.S, dtype=f32, device=device)
        self.neg = torch.empty(max_envs, 8, self.T, 3, dtype=f32, device=device)
        self.pos = torch.empty(max_envs, 8, self.T, 3, dtype=f32, device=device)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, config=None, max_envs: int = 64, **kwargs):
        """the reference's loader (navdp_policy.py:36-64): `NavDPNet.from_pretrained(path, config=NavDPModelConfig(model_cfg=...))` with
        `path` a directory holding pytorch_model.bin or a single state-dict file; model_cfg['il'] supplies memory_size / predict_size /
        temporal_depth / heads / token_dim, model_cfg['local_rank'] the device (`cuda:{local_rank}`, :74)."""
        import os

        from . import synthetic

        mc = getattr(config, "model_cfg", config) or {}
        if hasattr(mc, "model_dump"):
            mc = mc.model_dump()
        il = dict(mc.get("il", {}))
        cfg = dict(synthetic.NAVDPNET_CFG)
        for k in ("image_size", "memory_size", "predict_size", "temporal_depth", "heads", "token_dim"):
            if k in il:
                cfg[k] = il[k]
        f = os.path.join(pretrained_model_name_or_path, "pytorch_model.bin") if os.path.isdir(pretrained_model_name_or_path) else pretrained_model_name_or_path
        sd = torch.load(f, map_location="cpu", weights_only=True)
        missing = [k for k in synthetic.navdpnet_spec(cfg) if k not in sd]
        if missing:
            raise KeyError(f"NavDPNet checkpoint {f} lacks {len(missing)} parameters of the point-goal path, e.g. {missing[:4]}")
        return cls(sd, cfg, device=f"cuda:{int(mc.get('local_rank', 0))}", max_envs=max_envs)

    def eval(self):
        return self

    def encode_rgbd(self, B: int, images: torch.Tensor, depths: torch.Tensor):
        """RGBDBackbone.forward (navdp_backbone.py:248-286): tokens -> cond rows 4.. (+ cond_pos_embed[4:])."""
        M, Lc, D = self.M, self.Lc, self.D
        nt = (M + 1) * 256
        tok = self.former.tokens[: B * nt]
        self.rgb.forward(images.reshape(B * M, 224, 224, 3), self.vit_ws, tok, out_map=(M * 256, nt, 0),
                         pos=self.former.pe[: M * 256], mean=IMAGENET_MEAN, std=IMAGENET_STD)
        self.depth_vit.forward(depths.reshape(B, 224, 224, 1), self.vit_ws, tok, out_map=(256, nt, M * 256),
                               pos=self.former.pe[M * 256:])
        cond3 = self.cond[: B * Lc].view(B, Lc, D)
        self.former.forward(B, cond3[:, 4:, :], residual=self.cond_pos[4:])

    def predict_pointgoal_batch_action_vel(self, goal_point: torch.Tensor, input_images: torch.Tensor, input_depths: torch.Tensor,
                                           x_init: torch.Tensor, step_noise: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """goal_point f32 [B,3]; input_images f32 [B,M,224,224,3] in 0..1; input_depths f32 [B,1,224,224,1] metres;
        x_init f32 [B,S,T,3]; step_noise f32 [K,B,S,T,3]  ->  (negative, positive) f32 [B,8,T,3]."""
        return self._predict_batch_action_vel(goal_point, input_images, input_depths, x_init, step_noise)

    def predict_nogoal_batch_action_vel(self, input_images: torch.Tensor, input_depths: torch.Tensor, x_init: torch.Tensor,
                                        step_noise: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """the zero-goal sibling (navdp_policy.py:323-339): the three goal slots of the condition hold `zeros_like(rgbd_embed[:, 0:1])`
        instead of the embedded point goal (so only `cond_pos_embed` remains in them); sampler, critic and ranking are the same.
        Same tensor arguments as the point-goal call minus the goal."""
        return self._predict_batch_action_vel(None, input_images, input_depths, x_init, step_noise)

    def _predict_batch_action_vel(self, goal_point: Optional[torch.Tensor], input_images: torch.Tensor, input_depths: torch.Tensor,
                                  x_init: torch.Tensor, step_noise: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        B = input_images.shape[0]
        assert B <= self.b_max and (goal_point is None or goal_point.shape[0] == B)
        S, T, D, Lc = self.S, self.T, self.D, self.Lc
        self.encode_rgbd(B, input_images, input_depths)
        cond = self.cond[: B * Lc]
        for j in (1, 2, 3):  # the goal embedding (point goal, or zeros: table fill) fills the three goal slots (navdp_policy.py:162)
            if goal_point is None:
                ops.embed3(None, None, None, out=cond, pos=self.cond_pos[j:j + 1], rows=B, out_map=(1, Lc, j))
            else:
                ops.embed3(goal_point, self.pt_w, self.pt_b, out=cond, pos=self.cond_pos[j:j + 1], rows=B, out_map=(1, Lc, j))
        sample = self._denoise(B, x_init, step_noise)
        # critic (navdp_policy.py:172-185): no-goal condition with slots 0..3 masked -> the K/V of cond rows 4.. are reused,
        # rows 0..3 are excluded by kv_start = 4 (memory_mask), no causal mask.
        rows = B * S * T
        ops.embed3(sample, self.in_w, self.in_b, out=self.ws.x[:rows], pos=self.out_pos)
        for l, L in enumerate(self.layers):
            decoder_layer_prenorm(L, self.ws, B * S, T, self.kv[l][: B * Lc], B, Lc, self.heads, "gelu", causal=False, kv_start=4)
        ops.seqpool_head(self.ws.x[:rows], T, self.ln_w, self.ln_b, self.cr_w, self.cr_b, self.critic[: B * S], eps=1e-5)
        ops.select_traj(self.critic[: B * S].view(B, S), sample.view(B, S, T, 3), self.neg[:B], self.pos[:B], k=8, scale=0.25)
        return self.neg[:B], self.pos[:B]


class NavDPPolicyDAT(_NavDPBase):
    """MI355X engine behind `NavDP_Policy_DPT_CriticSum_DAT.predict_pointgoal_action_async` (internvla_n1/navdp.py:197-253) and, with
    use_async=False, `predict_pointgoal_action` (:255-289: the 'navdp' System-1 type - condition = [time, mean of the embedded VLM tokens],
    no RGB-D memory, no TokenCompressor)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: dict, device="cuda:0", max_envs: int = 64, n_query: int = 4, use_async: bool = True):
        device = torch.device(device)
        self.cfg = cfg
        M = cfg["memory_size"]
        self.M, self.nq_vlm, self.use_async = M, n_query, bool(use_async)
        Lc = (M * 16 if self.use_async else 0) + 2
        self._init_denoiser(state_dict, device, cfg, max_envs, Lc)
        sd, p = state_dict, "rgbd_encoder."
        bf, f32 = torch.bfloat16, torch.float32
        if self.use_async:
            self.rgb = DinoV2Encoder(sd, p + "rgb_model.", device)
            self.depth_vit = DinoV2Encoder(sd, p + "depth_model.", device)
            self.vit_ws = VitWorkspace(max_envs * M, device)
            self.former = _RGBDFormer(sd, p, device, M * 16, 2 * M * 256, max_envs, "former_query.weight", "former_pe.weight")
        self.cond_pos = sd["cond_pos_embed"][0].to(device=device, dtype=f32).contiguous()
        self.out_pos = sd["out_pos_embed"][0].to(device=device, dtype=f32).contiguous()
        self._set_time_tables(sd["cond_pos_embed"][0, 0])
        self.mlp = [(sd[f"vlm_embed_mlp.{i}.weight"].to(device=device, dtype=bf).contiguous(),
                     sd[f"vlm_embed_mlp.{i}.bias"].to(device=device, dtype=f32).contiguous()) for i in (0, 2, 4)]
        V = cfg["vlm_token_dim"]
        self.m0 = torch.empty(max_envs * n_query, V // 4, dtype=bf, device=device)
        self.m1 = torch.empty(max_envs * n_query, V // 8, dtype=bf, device=device)
        if not self.use_async:
            self.gc_tok = torch.empty(max_envs * n_query, self.D, dtype=bf, device=device)
            return
        # TokenCompressor (navdp_backbone.py:60-99): 1 learned query, 8 heads
        g = "goal_compressor."
        self.tok_pe = sd[g + "token_positional_encoding.position_embedding.weight"][:n_query].to(device=device, dtype=f32).contiguous()
        W, b = sd[g + "cross_attention.in_proj_weight"], sd[g + "cross_attention.in_proj_bias"]
        D = self.D
        q = (sd[g + "target_embedding.weight"] + sd[g + "query_positional_encoding.position_embedding.weight"][:1]).float()
        # the single query is input independent: q_proj = Wq q + bq is tabulated at load (host fp32)
        self.gc_q = (q @ W[:D].float().t() + b[:D].float()).to(device=device, dtype=bf).reshape(1, 1, 8, D // 8).contiguous()
        self.gc_kvw, self.gc_kvb = W[D:].to(device=device, dtype=bf).contiguous(), b[D:].to(device=device, dtype=f32).contiguous()
        self.gc_ow = sd[g + "cross_attention.out_proj.weight"].to(device=device, dtype=bf).contiguous()
        self.gc_ob = sd[g + "cross_attention.out_proj.bias"].to(device=device, dtype=f32).contiguous()
        self.gc_tok = torch.empty(max_envs * n_query, D, dtype=bf, device=device)
        self.gc_kv = torch.empty(max_envs * n_query, 2 * D, dtype=bf, device=device)
        self.gc_att = torch.empty(max_envs, D, dtype=bf, device=device)
        self.gc_qb = self.gc_q.expand(max_envs, 1, 8, D // 8).contiguous()

    def encode_rgbd(self, B: int, images: torch.Tensor, depths: torch.Tensor):
        """DAT_RGBD_Patch_Backbone.forward, version > 0 (navdp_backbone.py:151-202) -> cond rows 2.. (+ cond_pos_embed[2:])."""
        M, Lc, D = self.M, self.Lc, self.D
        nt = 2 * M * 256
        tok = self.former.tokens[: B * nt]
        self.rgb.forward(images.reshape(B * M, 224, 224, 3), self.vit_ws, tok, out_map=(M * 256, nt, 0),
                         pos=self.former.pe[: M * 256], mean=IMAGENET_MEAN_BF16, std=IMAGENET_STD_BF16)
        self.depth_vit.forward(depths.reshape(B * M, 224, 224, 1), self.vit_ws, tok, out_map=(M * 256, nt, M * 256),
                               pos=self.former.pe[M * 256:])
        cond3 = self.cond[: B * Lc].view(B, Lc, D)
        self.former.forward(B, cond3[:, 2:, :], residual=self.cond_pos[2:])

    def predict_pointgoal_action_async(self, vlm_tokens: torch.Tensor, input_images: torch.Tensor, input_depths: torch.Tensor,
                                       x_init: torch.Tensor, step_noise: torch.Tensor) -> torch.Tensor:
        """vlm_tokens bf16 [B,n_query,3584]; input_images [B,M,224,224,3] in 0..1; input_depths [B,M,224,224,1] metres
        (f32 or bf16); x_init f32 [B,S,T,3]; step_noise f32 [K,B,S,T,3]  ->  trajectories f32 [B,S,T,3]."""
        B, nq = vlm_tokens.shape[0], self.nq_vlm
        assert B <= self.b_max and vlm_tokens.shape[1] == nq
        D, Lc = self.D, self.Lc
        self.encode_rgbd(B, input_images, input_depths)
        # vlm_embed_mlp (navdp.py:94-100,237) + TokenCompressor -> cond row 1 (+ cond_pos_embed[1])
        rows = B * nq
        ops.linear(vlm_tokens.reshape(rows, -1), self.mlp[0][0], bias=self.mlp[0][1], act="relu", out=self.m0[:rows])
        ops.linear(self.m0[:rows], self.mlp[1][0], bias=self.mlp[1][1], act="relu", out=self.m1[:rows])
        ops.linear(self.m1[:rows].view(B, nq, -1), self.mlp[2][0], bias=self.mlp[2][1], residual=self.tok_pe,
                   out=self.gc_tok[:rows].view(B, nq, D), batched=True)
        ops.linear(self.gc_tok[:rows], self.gc_kvw, bias=self.gc_kvb, out=self.gc_kv[:rows])
        kv5 = self.gc_kv[:rows].view(B, nq, 2, 8, D // 8)
        ops.attention(self.gc_qb[:B], kv5[:, :, 0], kv5[:, :, 1], out=self.gc_att[:B].view(B, 1, 8, D // 8))
        cond3 = self.cond[: B * Lc].view(B, Lc, D)
        ops.linear(self.gc_att[:B].view(B, 1, D), self.gc_ow, bias=self.gc_ob, residual=self.cond_pos[1:2], out=cond3[:, 1:2, :],
                   batched=True)
        sample = self._denoise(B, x_init, step_noise)
        return sample.view(B, self.S, self.T, 3)


    def predict_pointgoal_action(self, vlm_tokens: torch.Tensor, x_init: torch.Tensor, step_noise: torch.Tensor) -> torch.Tensor:
        """the non-async 'navdp' System-1 (internvla_n1/navdp.py:255-289): vlm_embed = mean over the tokens of vlm_embed_mlp(vlm_tokens),
        condition = [time, vlm_embed] + cond_pos_embed[:2], num_train_timesteps DDPM steps. vlm_tokens bf16 [B, n_query, 3584]; x_init f32
        [B,S,T,3]; step_noise f32 [K,B,S,T,3] -> f32 [B,S,T,3]. Every env of the batch gets its own condition (the reference keeps only
        the first sample of a batch, :265-268 - it is only ever called with batch 1)."""
        assert not self.use_async, "this engine was built for the async head: use predict_pointgoal_action_async"
        B, nq = vlm_tokens.shape[0], self.nq_vlm
        assert B <= self.b_max and vlm_tokens.shape[1] == nq
        D, Lc = self.D, self.Lc
        rows = B * nq
        ops.linear(vlm_tokens.reshape(rows, -1), self.mlp[0][0], bias=self.mlp[0][1], act="relu", out=self.m0[:rows])
        ops.linear(self.m0[:rows], self.mlp[1][0], bias=self.mlp[1][1], act="relu", out=self.m1[:rows])
        ops.linear(self.m1[:rows], self.mlp[2][0], bias=self.mlp[2][1], out=self.gc_tok[:rows])
        cond3 = self.cond[: B * Lc].view(B, Lc, D)
        ops.pool_act(self.gc_tok[:rows], cond3[:, 1, :], T=nq, pos=self.cond_pos[1:2].contiguous())      # mean over tokens + cond_pos_embed[1]
        return self._denoise(B, x_init, step_noise).view(B, self.S, self.T, 3)


class NavDPModelConfig:
    """holder of `model_cfg` as the reference's NavDPModelConfig is used by its callers (navdp_policy.py:22-33: model_cfg with the
    'model', 'il' and 'local_rank' entries)."""
    model_type = "navdp"

    def __init__(self, model_cfg=None, **kwargs):
        self.model_cfg = model_cfg if model_cfg is not None else {}
        for k, v in kwargs.items():
            setattr(self, k, v)
